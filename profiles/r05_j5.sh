#!/bin/bash
# Round 5, job 5: (1) tests after the wide kernel's rows share the chain list's prefix; db2db with a 66 000-sample part, (2) A/B: one LDS atomic
# per lane in row_emit (c3part), k2_apply's window at c2, phase times of the wide kernel, each default twice (box noise), (3) c3gpu, (4) new2all at c5gpu
TAG=r05_j5
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=6 -k "random_forests or randomised_stress or second_level or many_samples or (baseline_sample and 10000) or patterns_that_touch or synthetic_databases or all2all_dense_bit_exact or db2db_bit_exact or db2db_with_more or pools_too_small or shards_sum" > $OUT/${TAG}_tests_sel.log 2>&1; tail -12 $OUT/${TAG}_tests_sel.log | cut -c1-200
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
ab c3_lane_a "--workload c3part" KMDB_ROW_LANE_ATOMICS=1
ab c3_b "--workload c3part" KMDB_X=0
ab c3_lane_b "--workload c3part" KMDB_ROW_LANE_ATOMICS=1
ab c3_prof "--workload c3part" KMDB_K1W_PROF=1
ab c2_a "" KMDB_X=0
ab c2_win32 "" KMDB_K2_WIN=32
ab c2_b "" KMDB_X=0
ab c2_win64 "" KMDB_K2_WIN=64
ab c2_win8 "" KMDB_K2_WIN=8
ab c2_prof "" KMDB_K1W_PROF=1
ab c3gpu "--workload c3gpu" KMDB_X=0
timeout 1200 python bench.py --mode new2all --workload c5gpu --steps 3 --warmup 1 > $OUT/${TAG}_c5gpu.json 2> $OUT/${TAG}_c5gpu.err; tail -4 $OUT/${TAG}_c5gpu.err | cut -c1-300
python - <<PY
import json
try:
    d=json.loads(open("$OUT/${TAG}_c5gpu.json").read().strip().splitlines()[-1]); print("c5gpu", round(d["ms_per_step"],3), d["roofline"]["frac"], d["config"].get("hashtable_slots"), d["config"].get("patterns"))
except Exception as e: print("c5gpu no line", e)
PY
ls $OUT | grep ${TAG} | wc -l
