#!/bin/bash
# Round-6 closing evidence on the round's final sources.  Run from the repo root:  bash profiles/r06_close.sh
# Every step with its own limit, the later ones skipped when the job's clock says so (gpurun_out/r06_close_clock.txt):
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of c2, c3part and c3gpu on THIS code
#      -> profiles/latest_traffic*.json (stamped with the sources' hash; the driver's bench run at the round's end replays them)
#   2. rocprofv3 kernel stats of c2 / c3part / c3gpu + the timeline of one C2 call
#   3. SQ counters of C2 (round 5's table had none: VERDICT round 5, next 2) and of c3part
#   4. the whole GPU suite, KMDB_REQUIRE_REF=1 (a missing reference build fails instead of weakening a test), -rs
#   5. the default bench line (C2 + extra.c3part + extra.c3gpu + the secondary rows), replaying that traffic
#   6. bench.py --driver node (the product's multi-GPU driver: 8 prefix shards of c3gpu's database on the one GPU; RCCL on a one-rank communicator)
#   7. the secondary modes' own lines — as far as the clock allows
TAG=r06_close
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
LIMIT=${CLOSE_LIMIT:-3300}          # seconds this job may take in all
clock() { echo "$1 done at ${SECONDS}s" | tee -a $OUT/${TAG}_clock.txt; }
left() { [ $((LIMIT - SECONDS)) -gt $1 ]; }
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
BENCH_ARGS="--no-extra" timeout 500 bash profiles/collect_counters.sh ${TAG}_c2_sq sq1 sq2 sq3 > $OUT/${TAG}_cc_c2sq.log 2>&1; clock "SQ counters c2"
BENCH_ARGS="--workload c3part" timeout 600 bash profiles/collect_counters.sh ${TAG}_c3part_sq sq1 sq2 sq3 > $OUT/${TAG}_cc_c3sq.log 2>&1; clock "SQ counters c3part"
KMDB_TEST_PHASES=$OUT/${TAG}_test_phases.txt timeout 1500 python -m pytest tests -m gpu -q -rs --durations=12 > $OUT/${TAG}_tests.log 2>&1; tail -24 $OUT/${TAG}_tests.log | cut -c1-200
clock "GPU suite"
left 420 && { KMDB_VERBOSE=1 timeout 900 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; grep -v "synth build" $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.err2; mv $OUT/${TAG}_bench.err2 $OUT/${TAG}_bench.err; clock "default bench line"; }
left 300 && { KMDB_NODE_FORCE_RCCL=1 timeout 600 python bench.py --gpus 1 --driver node --workload c3gpu --shards 8 --steps 5 --warmup 2 2> $OUT/${TAG}_node8.err > $OUT/${TAG}_node8.json; clock "node driver, 8 shards of c3gpu on one GPU"; }
for m in all2all-sp; do left 130 && timeout 300 python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json; done
left 200 && timeout 400 python bench.py --mode all2all-sp --workload c4sparse 2> $OUT/${TAG}_mode_all2all-sp_c4sparse.err > $OUT/${TAG}_mode_all2all-sp_c4sparse.json
left 300 && timeout 600 python bench.py --mode new2all --workload c5gpu --steps 3 --warmup 1 2> $OUT/${TAG}_mode_new2all_c5gpu.err > $OUT/${TAG}_mode_new2all_c5gpu.json
clock "secondary modes"
python - <<PY
import json
try:
    b=json.load(open("$OUT/${TAG}_bench.json"))
    print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], b["roofline"]["per_kernel_ms"], b["roofline"]["block_records_per_launch"], b["roofline"]["first_block_records_per_launch"])
    print("   ", {k:v for k,v in b["wall"].items() if k.startswith("frontend") or k.startswith("reference") or k.startswith("upload")}, b["cpu_baseline"].get("sweep"))
    for n in ("c3part","c3gpu"):
        e=b["extra"][n]; print(n, round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["roofline"]["traffic"], e["per_kernel_ms"], e["records"], e.get("records_applied_from_slices"), e.get("nodes_joined_per_tile"), e.get("rows_from_definition"), e.get("reference_match"), "upload", e.get("upload_s"))
    for n in ("new2all_c5part","db2db_parts"):
        e=b["extra"][n]; print(n, round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), (e.get("cpu_baseline") or {}).get("kind"), (e.get("cpu_baseline") or {}).get("seconds"), round(e["seconds_in_bench"],1))
except Exception as ex: print("bench line:", ex)
for m in ("node8","mode_all2all-sp","mode_all2all-sp_c4sparse","mode_new2all_c5gpu"):
    try:
        d=json.loads([l for l in open("$OUT/${TAG}_%s.json"%m).read().strip().splitlines() if l.startswith("{")][-1]); print(m, round(d["ms_per_step"],3), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("seconds"), d["config"].get("per_rank"))
    except Exception as ex: print(m, "no line", ex)
PY
tail -4 $OUT/${TAG}_c2_timeline.md
cat $OUT/${TAG}_clock.txt; cat $OUT/${TAG}_test_phases.txt
ls $OUT | grep ${TAG} | wc -l
