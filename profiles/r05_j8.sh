#!/bin/bash
# Round 5, job 8: the wide kernel without global loads inside its rounds (the topmost in-batch ancestor's (blocks, masks) come across the lanes);
# chunks per device atomic (KMDB_ARENA_GRAB 4 / 16); tests of the wide path; c3part / c2 twice each + profile
TAG=r05_j8
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=3 -k "random_forests or randomised_stress or second_level or many_samples or patterns_that_touch or synthetic_databases or pools_too_small" > $OUT/${TAG}_tests_sel.log 2>&1; tail -6 $OUT/${TAG}_tests_sel.log | cut -c1-200
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"), d["roofline"].get("nodes_joined_per_tile"))
except Exception as e: print("$name: no line", e)
PY
  grep -h "k1w phases\|k1w counts" $OUT/${TAG}_ab_$name.err | tail -2
}
ab c3_a "--workload c3part" KMDB_X=0
ab c3_g16_a "--workload c3part" KMDB_ARENA_GRAB=16
ab c3_b "--workload c3part" KMDB_X=0
ab c3_g16_b "--workload c3part" KMDB_ARENA_GRAB=16
ab c3_prof "--workload c3part" KMDB_K1W_PROF=1
ab c2_a "" KMDB_X=0
ab c2_g16 "" KMDB_ARENA_GRAB=16
ab c2_b "" KMDB_X=0
ls $OUT | grep ${TAG} | wc -l
