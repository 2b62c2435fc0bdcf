#!/bin/bash
# Round 5, job 14: rs_hist with 8 / 16 chunks' key words requested together (KMDB_RSH_UNROLL), K0 without the reservation scan in waves that
# need no pair slots; the new tests (few-streams record forms, protein goldens through the front-end)
TAG=r05_j14
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 -k "few_streams_record_forms or cli_byte or all2all_dense_bit_exact or random_forests or randomised_stress or second_level or pools_too_small or (baseline_sample and 10000)" > $OUT/${TAG}_tests_sel.log 2>&1; tail -9 $OUT/${TAG}_tests_sel.log | cut -c1-200
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
ab c3_u4_a "--workload c3part" KMDB_RSH_UNROLL=4
ab c3_u8_a "--workload c3part" KMDB_RSH_UNROLL=8
ab c3_u16_a "--workload c3part" KMDB_RSH_UNROLL=16
ab c3_u4_b "--workload c3part" KMDB_RSH_UNROLL=4
ab c3_u8_b "--workload c3part" KMDB_RSH_UNROLL=8
ab c3_u16_b "--workload c3part" KMDB_RSH_UNROLL=16
ab c2_a "" KMDB_X=0
ab c2_b "" KMDB_X=0
ls $OUT | grep ${TAG} | wc -l
