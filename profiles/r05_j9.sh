#!/bin/bash
# Round 5, job 9: wide kernel at 16 waves per CU (LDS trimmed), pool headroom rule; tests of the wide path and of the enlarge-and-repeat paths,
# c3part / c2 twice each, c3gpu
TAG=r05_j9
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=3 -k "random_forests or randomised_stress or second_level or many_samples or patterns_that_touch or synthetic_databases or pools_too_small or all2all_dense_bit_exact or (baseline_sample and 10000)" > $OUT/${TAG}_tests_sel.log 2>&1; tail -6 $OUT/${TAG}_tests_sel.log | cut -c1-200
for i in 1 2 3; do timeout 300 python -m pytest tests -m gpu -q -x -k "pools_too_small" > $OUT/${TAG}_tests_pools$i.log 2>&1; tail -1 $OUT/${TAG}_tests_pools$i.log; done
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"), d["roofline"].get("nodes_joined_per_tile"))
except Exception as e: print("$name: no line", e)
PY
}
ab c3_a "--workload c3part" KMDB_X=0
ab c3_b "--workload c3part" KMDB_X=0
ab c2_a "" KMDB_X=0
ab c2_b "" KMDB_X=0
ab c3gpu "--workload c3gpu" KMDB_X=0
BENCH_ARGS="--workload c3part --no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
ls $OUT | grep ${TAG} | wc -l
