#!/bin/bash
# Round 5, job 6: the wide kernel after its walks read one descriptor per node and its batch loads are requested together; rs_scatter at three
# workgroups per CU.  Tests of the wide path, then c3part / c2 twice each (box noise) and the kernel's phase times.
TAG=r05_j6
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=4 -k "random_forests or randomised_stress or second_level or many_samples or (baseline_sample and 10000) or patterns_that_touch or synthetic_databases or all2all_dense_bit_exact or pools_too_small" > $OUT/${TAG}_tests_sel.log 2>&1; tail -8 $OUT/${TAG}_tests_sel.log | cut -c1-200
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"), d["roofline"].get("nodes_joined_per_tile"))
except Exception as e: print("$name: no line", e)
PY
  grep -h "k1w phases" $OUT/${TAG}_ab_$name.err | tail -1
}
ab c3_a "--workload c3part" KMDB_X=0
ab c3_b "--workload c3part" KMDB_X=0
ab c3_prof "--workload c3part" KMDB_K1W_PROF=1
ab c2_a "" KMDB_X=0
ab c2_b "" KMDB_X=0
ab c2_prof "" KMDB_K1W_PROF=1
BENCH_ARGS="--workload c3part --no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
ls $OUT | grep ${TAG} | wc -l
