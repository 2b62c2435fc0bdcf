#!/bin/bash
# Round-4 closing evidence on the GPU box (run from the repo root):  bash profiles/r04_final.sh <tag>
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of c2 and c3part on THIS code -> profiles/latest_traffic*.json of the box's
#      copy, so that the bench line that follows replays numbers measured on the same sources
#   2. the default bench line (C2 + extra.c3part, reference + front-end end to end)   3. c3gpu (one GPU's share of configs[2])
#   4. all2all-sp line   5. the whole GPU test suite
TAG=${1:-r04_final}
OUT=$PWD/gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-extra" bash profiles/collect_counters.sh ${TAG}_c2 sq1 sq2 sq3 fetch write > $OUT/${TAG}_cc_c2.log 2>&1
BENCH_ARGS="--workload c3part" bash profiles/collect_counters.sh ${TAG}_c3part sq1 sq2 sq3 fetch write > $OUT/${TAG}_cc_c3.log 2>&1
[ -s $OUT/${TAG}_c2_traffic.json ] && cp $OUT/${TAG}_c2_traffic.json profiles/latest_traffic.json
[ -s $OUT/${TAG}_c3part_traffic.json ] && cp $OUT/${TAG}_c3part_traffic.json profiles/latest_traffic_c3part.json
python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; tail -2 $OUT/${TAG}_bench.err
KMDB_VERBOSE=1 python bench.py --workload c3gpu --no-cpu-baseline 2> $OUT/${TAG}_c3gpu_bench.err > $OUT/${TAG}_c3gpu_bench.json
grep -v "synth build" $OUT/${TAG}_c3gpu_bench.err > $OUT/${TAG}_c3gpu_bench.err2; mv $OUT/${TAG}_c3gpu_bench.err2 $OUT/${TAG}_c3gpu_bench.err
for m in all2all-sp new2all db2db; do python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json; done
BENCH_ARGS="--workload c3part --no-extra" bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
BENCH_ARGS="--no-extra" bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
BENCH_ARGS="--mode new2all" bash profiles/collect_profiles.sh ${TAG}_mode_new2all stats > $OUT/${TAG}_cp_n2a.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json")); e=b["extra"]["c3part"]
print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], b["roofline"]["per_kernel_ms"])
print("c3part", round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["roofline"]["traffic"], e["per_kernel_ms"])
c=json.load(open("$OUT/${TAG}_c3gpu_bench.json")); print("c3gpu", round(c["ms_per_step"],3), c["roofline"]["per_kernel_ms"])
for m in ("all2all-sp","new2all","db2db"):
    d=json.loads(open("$OUT/${TAG}_mode_%s.json"%m).read().strip().splitlines()[-1]); print(m, round(d["ms_per_step"],3), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("seconds"))
PY
ls $OUT | grep ${TAG} | head -40
