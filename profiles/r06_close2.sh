#!/bin/bash
# Round-6 closing evidence, second part, on the round's FINAL sources (after r06_close.sh: the many-streams record pool no longer runs out on the
# first call — r06_close's traffic of c3part / c3gpu held FIVE pipeline runs in "four calls" —, k2d_kernel counted by the summaries):
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of c2, c3part, c3gpu -> profiles/latest_traffic*.json (sources' hash)
#   2. rocprofv3 kernel stats of c2 / c3part / c3gpu + the timeline of one C2 call
#   3. the default bench line replaying that traffic
#   4. the parity tests of the touched paths
TAG=r06_close2
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
clock() { echo "$1 done at ${SECONDS}s" | tee -a $OUT/${TAG}_clock.txt; }
BENCH_ARGS="--no-extra" timeout 400 bash profiles/collect_counters.sh ${TAG}_c2 fetch write > $OUT/${TAG}_cc_c2.log 2>&1; clock "traffic c2"
BENCH_ARGS="--workload c3part" timeout 500 bash profiles/collect_counters.sh ${TAG}_c3part fetch write > $OUT/${TAG}_cc_c3.log 2>&1; clock "traffic c3part"
BENCH_ARGS="--workload c3gpu" timeout 600 bash profiles/collect_counters.sh ${TAG}_c3gpu fetch write > $OUT/${TAG}_cc_c3gpu.log 2>&1; clock "traffic c3gpu"
[ -s $OUT/${TAG}_c2_traffic.json ] && cp $OUT/${TAG}_c2_traffic.json profiles/latest_traffic.json
[ -s $OUT/${TAG}_c3part_traffic.json ] && cp $OUT/${TAG}_c3part_traffic.json profiles/latest_traffic_c3part.json
[ -s $OUT/${TAG}_c3gpu_traffic.json ] && cp $OUT/${TAG}_c3gpu_traffic.json profiles/latest_traffic_c3gpu.json
BENCH_ARGS="--no-extra" timeout 300 bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
for f in $(find /tmp/prof_stats -name '*kernel_trace.csv'); do
  if grep -q "k0_decode_kernel" $f; then python profiles/timeline.py $f $OUT/${TAG}_c2_timeline.md > /dev/null; fi
done
BENCH_ARGS="--workload c3part --no-extra" timeout 300 bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
BENCH_ARGS="--workload c3gpu --no-extra" timeout 400 bash profiles/collect_profiles.sh ${TAG}_c3gpu stats > $OUT/${TAG}_cp_c3gpu.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
clock "kernel stats c2, c3part, c3gpu"
timeout 900 python -m pytest tests -m gpu -q -rs -k "all2all_dense_bit_exact or random_forests or degenerate or synthetic_databases or sparse or second_level or pools_too_small or 10000-50-400 or many_samples or touch_every_block or node_driver or cli_byte" > $OUT/${TAG}_tests.log 2>&1; tail -5 $OUT/${TAG}_tests.log | cut -c1-200
clock "tests"
KMDB_VERBOSE=1 timeout 900 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; grep -v "synth build" $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.err2; mv $OUT/${TAG}_bench.err2 $OUT/${TAG}_bench.err; clock "default bench line"
python - <<PY
import json
try:
    b=json.load(open("$OUT/${TAG}_bench.json"))
    print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], b["roofline"]["per_kernel_ms"], b["roofline"]["block_records_per_launch"], b["roofline"]["first_block_records_per_launch"])
    print("   ", {k:v for k,v in b["wall"].items() if k.startswith("frontend") or k.startswith("reference") or k.startswith("upload") or k.startswith("cold")}, b["cpu_baseline"].get("sweep"))
    for n in ("c3part","c3gpu"):
        e=b["extra"][n]; print(n, round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["roofline"]["traffic"], e["per_kernel_ms"], e["records"], e.get("records_applied_from_slices"), e.get("nodes_joined_per_tile"), e.get("rows_from_definition"), e.get("reference_match"), "upload", e.get("upload_s"))
    for n in ("new2all_c5part","db2db_parts"):
        e=b["extra"][n]; print(n, round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), (e.get("cpu_baseline") or {}).get("kind"), (e.get("cpu_baseline") or {}).get("seconds"), round(e["seconds_in_bench"],1))
except Exception as ex: print("bench line:", ex)
PY
grep -n "too small\|enlarged\|record pool:" $OUT/${TAG}_bench.err | head
tail -3 $OUT/${TAG}_c2_timeline.md
cat $OUT/${TAG}_clock.txt
