#!/bin/bash
# Round 6, job 3: the narrow kernel applies its first-block records itself (k1n_kernel<true>: bit transpose + byte spreading + six MFMAs per batch
# of 64 nodes, tile (X, X) in 48 accumulator registers for the whole slice, one write-back) instead of writing them to stream chunks for the
# apply kernel on the side stream.  Parity of everything that touches the all2all path, then A/B: KMDB_K1N_DIRECT=0 (round 5's path) /
# 1 at three waves per SIMD (no spills) / 1 at four (84 B of scratch per lane), C2 twice, c3part once each.
TAG=r06_j3
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 -k "all2all or synthetic_databases or random_forests or randomised_stress or pools_too_small or degenerate or second_level or many_samples or touch_every_block or 10000-50-400 or few_streams or shards_sum or upload_shards or sparse" > $OUT/${TAG}_tests_sel.log 2>&1; tail -9 $OUT/${TAG}_tests_sel.log | cut -c1-200
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
ab c2_chunks_a "" KMDB_K1N_DIRECT=0
ab c2_direct3_a "" KMDB_K1N_DIRECT=1
ab c2_direct4_a "" KMDB_K1N_DIRECT=1 KMDB_K1N_MINW=4
ab c2_chunks_b "" KMDB_K1N_DIRECT=0
ab c2_direct3_b "" KMDB_K1N_DIRECT=1
ab c2_direct4_b "" KMDB_K1N_DIRECT=1 KMDB_K1N_MINW=4
ab c3_chunks "--workload c3part" KMDB_K1N_DIRECT=0
ab c3_direct3 "--workload c3part" KMDB_K1N_DIRECT=1
ab c3_direct4 "--workload c3part" KMDB_K1N_DIRECT=1 KMDB_K1N_MINW=4
ls $OUT | grep ${TAG} | wc -l
