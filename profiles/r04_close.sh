#!/bin/bash
# Round-4 closing evidence, second edition (sources after r04_final changed: 20-bit sample ids, new2all node records, the front-end's reader /
# teardown work).  Run from the repo root:  bash profiles/r04_close.sh
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes) of c2 and c3part on THIS code -> profiles/latest_traffic*.json of the box's copy
#   2. default bench line (C2 + extra.c3part; reference and front-end end to end)   3. c3gpu   4. the secondary modes
#   5. rocprofv3 kernel stats of c2 / c3part + the timeline of one C2 call   6. the whole GPU test suite
TAG=r04_close
OUT=$PWD/gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-extra" bash profiles/collect_counters.sh ${TAG}_c2 fetch write > $OUT/${TAG}_cc_c2.log 2>&1
BENCH_ARGS="--workload c3part" bash profiles/collect_counters.sh ${TAG}_c3part fetch write > $OUT/${TAG}_cc_c3.log 2>&1
[ -s $OUT/${TAG}_c2_traffic.json ] && cp $OUT/${TAG}_c2_traffic.json profiles/latest_traffic.json
[ -s $OUT/${TAG}_c3part_traffic.json ] && cp $OUT/${TAG}_c3part_traffic.json profiles/latest_traffic_c3part.json
python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; grep -v "synth build" $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.err2; mv $OUT/${TAG}_bench.err2 $OUT/${TAG}_bench.err
KMDB_VERBOSE=1 python bench.py --workload c3gpu --no-cpu-baseline 2> $OUT/${TAG}_c3gpu_bench.err > $OUT/${TAG}_c3gpu_bench.json
grep -v "synth build" $OUT/${TAG}_c3gpu_bench.err > $OUT/${TAG}_c3gpu_bench.err2; mv $OUT/${TAG}_c3gpu_bench.err2 $OUT/${TAG}_c3gpu_bench.err
for m in all2all-sp new2all db2db; do python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json; done
BENCH_ARGS="--workload c3part --no-extra" bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
BENCH_ARGS="--no-extra" bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
for f in $(find /tmp/prof_stats -name '*kernel_trace.csv'); do
  if grep -q "k0_decode_kernel" $f; then python profiles/timeline.py $f $OUT/${TAG}_c2_timeline.md > /dev/null; fi
done
rm -f $OUT/*_kernel_stats_all.csv
python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json")); e=b["extra"]["c3part"]
print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], b["roofline"]["per_kernel_ms"])
print("   ", {k:v for k,v in b["wall"].items() if k.startswith("frontend") or k.startswith("reference")})
print("c3part", round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["roofline"]["traffic"], e["per_kernel_ms"])
print("   ", {k:v for k,v in e.items() if k.startswith("frontend") or k.startswith("reference")})
c=json.load(open("$OUT/${TAG}_c3gpu_bench.json")); print("c3gpu", round(c["ms_per_step"],3), c["roofline"]["per_kernel_ms"])
for m in ("all2all-sp","new2all","db2db"):
    d=json.loads(open("$OUT/${TAG}_mode_%s.json"%m).read().strip().splitlines()[-1]); print(m, round(d["ms_per_step"],3), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("seconds"))
PY
tail -4 $OUT/${TAG}_c2_timeline.md
ls $OUT | grep ${TAG} | head -40
