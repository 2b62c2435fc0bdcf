#!/bin/bash
# Round 5, job 3: (1) the apply step by stream jobs after its fix (job 2: the stream starts were read from the wrong offsets table) — the tests that
# take the many-streams path — (2) A/B: KMDB_SHORT_IDS (local lists of up to 32 / 48 / 64 ids decoded by the short launch) with the two-deep fetch of
# the narrow kernel, c2 and c3part, (3) phase times of the wide kernel (KMDB_K1W_PROF), (4) parity of the kept variants
TAG=r05_j3
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x --durations=5 -k "second_level or many_samples or (baseline_sample and 10000) or patterns_that_touch or pools_too_small or node_driver or upload_shards or integration_glue" > $OUT/${TAG}_tests_sel.log 2>&1; tail -8 $OUT/${TAG}_tests_sel.log
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"))
except Exception as e: print("$name: no line", e)
PY
  grep -h "k1w phases" $OUT/${TAG}_ab_$name.err | tail -1
}
ab c2_base "" KMDB_K1NV=0
ab c2_n1 "" KMDB_K1NV=1
ab c2_n1_s48 "" KMDB_K1NV=1 KMDB_SHORT_IDS=48
ab c2_n1_s64 "" KMDB_K1NV=1 KMDB_SHORT_IDS=64
ab c2_prof "" KMDB_K1NV=1 KMDB_K1W_PROF=1
ab c3_base "--workload c3part" KMDB_K1NV=0
ab c3_n1 "--workload c3part" KMDB_K1NV=1
ab c3_n1_s64 "--workload c3part" KMDB_K1NV=1 KMDB_SHORT_IDS=64
ab c3_prof "--workload c3part" KMDB_K1NV=1 KMDB_K1W_PROF=1
ab c3_l2off "--workload c3part" KMDB_K1NV=1 KMDB_L2=0
KMDB_K1NV=1 KMDB_SHORT_IDS=64 timeout 600 python -m pytest tests -m gpu -q -x -k "all2all_dense_bit_exact or random_forests or synthetic_databases or second_level or many_samples" > $OUT/${TAG}_tests_var.log 2>&1; tail -3 $OUT/${TAG}_tests_var.log
BENCH_ARGS="--workload c3part --no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
ls $OUT | grep ${TAG} | head -40
