#!/bin/bash
# Round 6, job 15: the wide kernel split in two (KMDB_K1W_SPLIT=1): k1w_kernel<true> writes its emitting nodes' lists to an entry pool (10 B per entry),
# k1e_kernel emits the records from them (no chain tables or rows: 70 VGPRs, ~9 KB of LDS per wave).  Parity of the wide-node paths with the split on
# (incl. pools of 1 %: the entry pool's doubling), then A/B at C2 (twice), c3part, c3gpu.
TAG=r06_j15
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
KMDB_K1W_SPLIT=1 timeout 1500 python -m pytest tests -m gpu -q -x -k "all2all or synthetic_databases or random_forests or randomised_stress or pools_too_small or degenerate or second_level or many_samples or touch_every_block or 10000-50-400 or few_streams or upload_shards or sparse or more_than_65535" > $OUT/${TAG}_tests_sel.log 2>&1; tail -5 $OUT/${TAG}_tests_sel.log | cut -c1-250
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" KMDB_VERBOSE=1 timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"]["block_records_per_launch"])
except Exception as e: print("$name: no line", e)
PY
  grep -h "split wide kernel\|entry pool" $OUT/${TAG}_ab_$name.err | head -3 | cut -c1-220
}
ab c2_whole_a "" KMDB_K1W_SPLIT=0
ab c2_split_a "" KMDB_K1W_SPLIT=1
ab c2_whole_b "" KMDB_K1W_SPLIT=0
ab c2_split_b "" KMDB_K1W_SPLIT=1
ab c3_whole "--workload c3part" KMDB_K1W_SPLIT=0
ab c3_split "--workload c3part" KMDB_K1W_SPLIT=1
ab c3gpu_whole "--workload c3gpu" KMDB_K1W_SPLIT=0
ab c3gpu_split "--workload c3gpu" KMDB_K1W_SPLIT=1
