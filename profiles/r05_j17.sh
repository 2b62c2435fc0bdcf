#!/bin/bash
# Round 5, job 17: the pools' initialisation (ten fills, 130 us at C2) on the side stream next to the decode launches, the narrow kernel
# waits for it (KMDB_INIT_SIDE=0: in front of the decode as before).  Parity of the paths it touches, C2 A/B twice, c3part once.
TAG=r05_j17
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests -m gpu -q -x --durations=5 -k "all2all_dense_bit_exact or random_forests or pools_too_small or shards_sum or second_level or few_streams_record_forms or degenerate or upload_shards" > $OUT/${TAG}_tests_sel.log 2>&1; tail -9 $OUT/${TAG}_tests_sel.log | cut -c1-200
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
ab c2_front_a "" KMDB_INIT_SIDE=0
ab c2_side_a "" KMDB_INIT_SIDE=1
ab c2_front_b "" KMDB_INIT_SIDE=0
ab c2_side_b "" KMDB_INIT_SIDE=1
ab c3_side "--workload c3part" KMDB_INIT_SIDE=1
ls $OUT | grep ${TAG} | wc -l
