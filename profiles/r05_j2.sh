#!/bin/bash
# Round 5, job 2: (1) tests of what changed since job 1 — apply step by stream jobs, host-planned shard uploads, node driver — (2) A/B of the
# decode / narrow experiments at c2 and c3part (KMDB_K0V: short nodes without the re-deal, 1 / 2 / 4 nodes per thread; KMDB_K1NV: two-deep fetch),
# (3) parity of the variants, (4) kernel stats + WRITE_SIZE of c3part
TAG=r05_j2
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --durations=8 -k "second_level or randomised_stress or node_driver or upload_shards or sharded or integration_glue or cli_byte or pools_too_small or random_forests or many_samples or (baseline_sample and 10000) or patterns_that_touch or db2db_bit_exact or all2all_dense_bit_exact" > $OUT/${TAG}_tests_sel.log 2>&1; tail -14 $OUT/${TAG}_tests_sel.log
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()})
except Exception as e: print("$name: no line", e)
PY
}
ab c2_k0v0_n0 "" KMDB_K0V=0 KMDB_K1NV=0
ab c2_k0v1_n0 "" KMDB_K0V=1 KMDB_K1NV=0
ab c2_k0v2_n0 "" KMDB_K0V=2 KMDB_K1NV=0
ab c2_k0v4_n0 "" KMDB_K0V=4 KMDB_K1NV=0
ab c2_k0v0_n1 "" KMDB_K0V=0 KMDB_K1NV=1
ab c2_k0v2_n1 "" KMDB_K0V=2 KMDB_K1NV=1
ab c3_k0v0_n0 "--workload c3part" KMDB_K0V=0 KMDB_K1NV=0
ab c3_k0v2_n1 "--workload c3part" KMDB_K0V=2 KMDB_K1NV=1
KMDB_K0V=2 KMDB_K1NV=1 timeout 600 python -m pytest tests -m gpu -q -x -k "all2all_dense_bit_exact or random_forests or synthetic_databases or second_level" > $OUT/${TAG}_tests_var.log 2>&1; tail -3 $OUT/${TAG}_tests_var.log
KMDB_K0V=4 KMDB_K1NV=0 timeout 600 python -m pytest tests -m gpu -q -x -k "all2all_dense_bit_exact or random_forests" > $OUT/${TAG}_tests_var4.log 2>&1; tail -3 $OUT/${TAG}_tests_var4.log
BENCH_ARGS="--workload c3part --no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
BENCH_ARGS="--workload c3part --no-extra" timeout 900 bash profiles/collect_counters.sh ${TAG}_c3part write > $OUT/${TAG}_cc_c3.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
ls $OUT | grep ${TAG} | head -40
