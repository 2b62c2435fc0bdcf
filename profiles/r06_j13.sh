#!/bin/bash
# Round 6, job 13: the slices' apply kernel started with the wide kernel (default now) instead of right behind the narrow kernel (KMDB_K2D_EARLY=1: the
# wide list's small kernels — scan, expand, root paths of the runs — then run beside it and take 0.28 ms at C2); slices of 4096 nodes (KMDB_NSEG).
TAG=r06_j13
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "all2all_dense_bit_exact or random_forests or degenerate or synthetic_databases or sparse_bit_exact or second_level or pools_too_small or 10000-50-400" > $OUT/${TAG}_tests_sel.log 2>&1; tail -3 $OUT/${TAG}_tests_sel.log | cut -c1-200
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()})
except Exception as e: print("$name: no line", e)
PY
}
ab c2_early_a "" KMDB_K2D_EARLY=1
ab c2_late_a "" KMDB_K2D_EARLY=0
ab c2_late_nseg4096_a "" KMDB_K2D_EARLY=0 KMDB_NSEG=4096
ab c2_early_b "" KMDB_K2D_EARLY=1
ab c2_late_b "" KMDB_K2D_EARLY=0
ab c2_late_nseg4096_b "" KMDB_K2D_EARLY=0 KMDB_NSEG=4096
ab c3_early "--workload c3part" KMDB_K2D_EARLY=1
ab c3_late "--workload c3part" KMDB_K2D_EARLY=0
ab c3_late_nseg4096 "--workload c3part" KMDB_K2D_EARLY=0 KMDB_NSEG=4096
