#!/bin/bash
# Round 5, job 1: the second level above the block records landed in csrc/ (profiles/r04_level2_integration.diff applied).
#   1. its own tests (forced thresholds 11 / 24, doubling path, slices, prefix shards) + the randomised stress with thresholds
#   2. default bench line (C2 + extra.c3part, reference beside it)   3. PMC traffic + SQ counters of c3part   4. kernel stats c3part / c2
#   5. the whole GPU suite with durations
TAG=r05_j1
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "second_level or randomised_stress" --durations=5 > $OUT/${TAG}_tests_l2.log 2>&1; tail -5 $OUT/${TAG}_tests_l2.log
KMDB_VERBOSE=1 timeout 900 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; grep -v "synth build" $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.err2; mv $OUT/${TAG}_bench.err2 $OUT/${TAG}_bench.err
BENCH_ARGS="--workload c3part" timeout 900 bash profiles/collect_counters.sh ${TAG}_c3part fetch write sq1 sq2 sq3 > $OUT/${TAG}_cc_c3.log 2>&1
BENCH_ARGS="--workload c3part --no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
BENCH_ARGS="--no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
python - <<PY
import json
try:
    b=json.load(open("$OUT/${TAG}_bench.json")); e=b["extra"]["c3part"]
    print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], b["roofline"]["per_kernel_ms"])
    print("c3part", round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["roofline"]["traffic"], e["per_kernel_ms"])
except Exception as ex: print("bench line:", ex)
PY
timeout 1500 python -m pytest tests -m gpu -q --durations=60 > $OUT/${TAG}_tests.log 2>&1; tail -70 $OUT/${TAG}_tests.log
ls $OUT | grep ${TAG} | head -40
