#!/bin/bash
# Round 6, job 5: the narrow kernel's first-block records compacted per slice in DFS order and applied by k2d_kernel on the side stream
# (KMDB_K1N_MODE=2) against round 5's stream chunks + k2_apply_kernel (0) and the step inside the narrow kernel (1).  Parity of everything
# on the all2all path in mode 2 (default) and a subset in mode 1, then the A/B: C2 twice, c3part once per mode.
TAG=r06_j5
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 -k "all2all or synthetic_databases or random_forests or randomised_stress or pools_too_small or degenerate or second_level or many_samples or touch_every_block or 10000-50-400 or few_streams or shards_sum or upload_shards or sparse or protein or node_driver or cli_byte" > $OUT/${TAG}_tests_sel.log 2>&1; tail -9 $OUT/${TAG}_tests_sel.log | cut -c1-220
KMDB_K1N_MODE=1 timeout 900 python -m pytest tests -m gpu -q -x -k "all2all_dense_bit_exact or random_forests or second_level or degenerate or pools_too_small or sparse_bit_exact" > $OUT/${TAG}_tests_mode1.log 2>&1; tail -3 $OUT/${TAG}_tests_mode1.log | cut -c1-220
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"))
except Exception as e: print("$name: no line", e)
PY
}
ab c2_mode0_a "" KMDB_K1N_MODE=0
ab c2_mode2_a "" KMDB_K1N_MODE=2
ab c2_mode1_a "" KMDB_K1N_MODE=1
ab c2_mode0_b "" KMDB_K1N_MODE=0
ab c2_mode2_b "" KMDB_K1N_MODE=2
ab c2_mode2_s1 "" KMDB_K1N_MODE=2 KMDB_K2D_SLICES=1
ab c2_mode2_s4 "" KMDB_K1N_MODE=2 KMDB_K2D_SLICES=4
ab c3_mode0 "--workload c3part" KMDB_K1N_MODE=0
ab c3_mode2 "--workload c3part" KMDB_K1N_MODE=2
ab c3_mode1 "--workload c3part" KMDB_K1N_MODE=1
ls $OUT | grep ${TAG} | wc -l
