#!/bin/bash
# Round 6, job 10: the short decode launch re-deals the nodes of a window of 256 by work; windows of 512 / 1024 nodes (KMDB_K0_BLOCK) make the waves
# more uniform (active lanes 42 % in round 5's counters).  Parity of the decode-heavy tests at 1024, then A/B at C2 (twice) and c3part.
TAG=r06_j10
OUT=$PWD/gpurun_out
mkdir -p $OUT
KMDB_K0_BLOCK=1024 timeout 900 python -m pytest tests -m gpu -q -x -k "all2all_dense_bit_exact or random_forests or degenerate or synthetic_databases" > $OUT/${TAG}_tests_sel.log 2>&1; tail -3 $OUT/${TAG}_tests_sel.log | cut -c1-200
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
ab c2_256_a "" KMDB_K0_BLOCK=256
ab c2_512_a "" KMDB_K0_BLOCK=512
ab c2_1024_a "" KMDB_K0_BLOCK=1024
ab c2_256_b "" KMDB_K0_BLOCK=256
ab c2_512_b "" KMDB_K0_BLOCK=512
ab c2_1024_b "" KMDB_K0_BLOCK=1024
ab c3_256 "--workload c3part" KMDB_K0_BLOCK=256
ab c3_1024 "--workload c3part" KMDB_K0_BLOCK=1024
