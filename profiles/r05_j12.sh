#!/bin/bash
# Round 5, job 12: few-streams path with packed 16-byte records (cs_scatter leaves the key words behind) and the apply step by stream jobs
# (k2_jobs_kernel instead of k2_sorted_kernel's windows): parity tests of the paths it touches, then C2 A/B against the unpacked / windowed forms
TAG=r05_j12
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=5 -k "all2all_dense_bit_exact or synthetic_databases or random_forests or randomised_stress or pools_too_small or degenerate or all2all_sparse or cli_byte or upload_shards or second_level or many_samples or heavy or weights" > $OUT/${TAG}_tests_sel.log 2>&1; tail -9 $OUT/${TAG}_tests_sel.log | cut -c1-200
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
ab c2_packed_a "" KMDB_X=0
ab c2_unpacked_jobs_a "" KMDB_REC_PACKED=0
ab c2_unpacked_windows_a "" KMDB_REC_PACKED=0 KMDB_K2_WINDOWS=1
ab c2_packed_b "" KMDB_X=0
ab c2_unpacked_jobs_b "" KMDB_REC_PACKED=0
ab c2_unpacked_windows_b "" KMDB_REC_PACKED=0 KMDB_K2_WINDOWS=1
BENCH_ARGS="--no-extra" timeout 600 bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
ls $OUT | grep ${TAG} | wc -l
