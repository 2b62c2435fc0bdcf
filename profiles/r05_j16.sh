#!/bin/bash
# Round 5, job 16: the second level's kernels (l2_ranks / l2_offsets / l2_lists / l2_join_apply) on a side stream of their own next to the
# sort inside the rows (both chains need only the wide kernel's output, both add into M by atomics).  Parity of the paths it touches,
# then c3part A/B: behind the wide kernel as before (KMDB_L2_SIDE=0) / on the side stream / side stream with the threshold at 16 and 12
# blocks (more nodes leave the record stream for the chain that now runs in the sort's shadow); c3gpu once.
TAG=r05_j16
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --durations=5 -k "second_level or 10000-50-400 or sparse_scans or pools_too_small or many_samples or all2all_dense_bit_exact or touch_every_block" > $OUT/${TAG}_tests_sel.log 2>&1; tail -9 $OUT/${TAG}_tests_sel.log | cut -c1-200
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"))
except Exception as e: print("$name: no line", e)
PY
}
ab c3_serial_a "--workload c3part" KMDB_L2_SIDE=0
ab c3_side_a "--workload c3part" KMDB_L2_SIDE=1
ab c3_side16_a "--workload c3part" KMDB_L2_SIDE=1 KMDB_L2_MIN=16
ab c3_side12_a "--workload c3part" KMDB_L2_SIDE=1 KMDB_L2_MIN=12
ab c3_serial_b "--workload c3part" KMDB_L2_SIDE=0
ab c3_side_b "--workload c3part" KMDB_L2_SIDE=1
ab c3_side16_b "--workload c3part" KMDB_L2_SIDE=1 KMDB_L2_MIN=16
ab c3_side12_b "--workload c3part" KMDB_L2_SIDE=1 KMDB_L2_MIN=12
ab c3gpu_side "--workload c3gpu" KMDB_L2_SIDE=1
ls $OUT | grep ${TAG} | wc -l
