#!/bin/bash
# Round-5 closing evidence on the round's final sources.  Run from the repo root:  bash profiles/r05_close.sh
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) + SQ counters of c2 and c3part, traffic of c3gpu, on THIS code
#      -> profiles/latest_traffic*.json of the box's copy (stamped with the sources' hash)
#   2. the default bench line (C2 + extra.c3part + extra.c3gpu; reference swept at full size; front-end end to end), replaying that traffic
#   3. the secondary modes   4. rocprofv3 kernel stats of c2 / c3part / c3gpu + the timeline of one C2 call   5. the whole GPU test suite
TAG=r05_close
OUT=$PWD/gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-extra" timeout 900 bash profiles/collect_counters.sh ${TAG}_c2 fetch write sq1 sq2 sq3 > $OUT/${TAG}_cc_c2.log 2>&1
BENCH_ARGS="--workload c3part" timeout 2100 bash profiles/collect_counters.sh ${TAG}_c3part fetch write sq1 sq2 sq3 > $OUT/${TAG}_cc_c3.log 2>&1
BENCH_ARGS="--workload c3gpu" timeout 1500 bash profiles/collect_counters.sh ${TAG}_c3gpu fetch write > $OUT/${TAG}_cc_c3gpu.log 2>&1
[ -s $OUT/${TAG}_c2_traffic.json ] && cp $OUT/${TAG}_c2_traffic.json profiles/latest_traffic.json
[ -s $OUT/${TAG}_c3part_traffic.json ] && cp $OUT/${TAG}_c3part_traffic.json profiles/latest_traffic_c3part.json
[ -s $OUT/${TAG}_c3gpu_traffic.json ] && cp $OUT/${TAG}_c3gpu_traffic.json profiles/latest_traffic_c3gpu.json
KMDB_VERBOSE=1 timeout 1500 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; grep -v "synth build" $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.err2; mv $OUT/${TAG}_bench.err2 $OUT/${TAG}_bench.err
for m in all2all-sp new2all db2db; do timeout 900 python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json; done
timeout 900 python bench.py --mode all2all-sp --workload c4sparse 2> $OUT/${TAG}_mode_all2all-sp_c4sparse.err > $OUT/${TAG}_mode_all2all-sp_c4sparse.json
timeout 1200 python bench.py --mode new2all --workload c5gpu --steps 3 --warmup 1 2> $OUT/${TAG}_mode_new2all_c5gpu.err > $OUT/${TAG}_mode_new2all_c5gpu.json
BENCH_ARGS="--workload c3gpu --no-extra" timeout 900 bash profiles/collect_profiles.sh ${TAG}_c3gpu stats > $OUT/${TAG}_cp_c3gpu.log 2>&1
BENCH_ARGS="--workload c3part --no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
BENCH_ARGS="--no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
for f in $(find /tmp/prof_stats -name '*kernel_trace.csv'); do
  if grep -q "k0_decode_kernel" $f; then python profiles/timeline.py $f $OUT/${TAG}_c2_timeline.md > /dev/null; fi
done
rm -f $OUT/*_kernel_stats_all.csv
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/${TAG}_tests.log 2>&1; tail -22 $OUT/${TAG}_tests.log | cut -c1-200
python - <<PY
import json
try:
    b=json.load(open("$OUT/${TAG}_bench.json"))
    print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], b["roofline"]["per_kernel_ms"])
    print("   ", {k:v for k,v in b["wall"].items() if k.startswith("frontend") or k.startswith("reference")}, b["cpu_baseline"].get("sweep"))
    for n in ("c3part","c3gpu"):
        e=b["extra"][n]; print(n, round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["roofline"]["traffic"], e["per_kernel_ms"], e["records"], e.get("nodes_joined_per_tile"), e.get("rows_from_definition"), e.get("reference_match"))
except Exception as ex: print("bench line:", ex)
for m in ("all2all-sp","new2all","db2db","all2all-sp_c4sparse","new2all_c5gpu"):
    try:
        d=json.loads(open("$OUT/${TAG}_mode_%s.json"%m).read().strip().splitlines()[-1]); print(m, round(d["ms_per_step"],3), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("seconds"))
    except Exception as ex: print(m, "no line", ex)
PY
tail -4 $OUT/${TAG}_c2_timeline.md
ls $OUT | grep ${TAG} | wc -l
