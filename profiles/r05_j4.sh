#!/bin/bash
# Round 5, job 4: (1) tests of the round's changes after the fixes (stream jobs clamped to the sorted arrays, packed records, tile-skipping sparse
# compaction, db2db beyond 65 535 samples is not in yet), (2) A/B at c3part: packed 16-byte records, second-level threshold, short/long threshold;
# c2: short/long threshold, (3) parity with the variants that look like keepers, (4) all2all-sp on sparse data (c4sparse) with / without tile skipping
TAG=r05_j4
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 -k "second_level or many_samples or (baseline_sample and 10000) or patterns_that_touch or pools_too_small or node_driver or upload_shards or integration_glue or randomised_stress or db2db_bit_exact or db2db_synthetic or sparse or random_forests" > $OUT/${TAG}_tests_sel.log 2>&1; tail -14 $OUT/${TAG}_tests_sel.log | cut -c1-200
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 300 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"), d["roofline"].get("nodes_joined_per_tile"))
except Exception as e: print("$name: no line", e)
PY
}
ab c3_n1 "--workload c3part" KMDB_K1NV=1
ab c3_n1_unpacked "--workload c3part" KMDB_K1NV=1 KMDB_REC_PACKED=0
ab c3_n1_s48 "--workload c3part" KMDB_K1NV=1 KMDB_SHORT_IDS=48
ab c3_n1_l2_16 "--workload c3part" KMDB_K1NV=1 KMDB_L2_MIN=16
ab c3_n1_l2_32 "--workload c3part" KMDB_K1NV=1 KMDB_L2_MIN=32
ab c3_n1_l2_40 "--workload c3part" KMDB_K1NV=1 KMDB_L2_MIN=40
ab c2_n1 "" KMDB_K1NV=1
ab c2_n1_s40 "" KMDB_K1NV=1 KMDB_SHORT_IDS=40
ab c2_n1_s48 "" KMDB_K1NV=1 KMDB_SHORT_IDS=48
ab c2_n1_s56 "" KMDB_K1NV=1 KMDB_SHORT_IDS=56
KMDB_K1NV=1 KMDB_SHORT_IDS=48 timeout 600 python -m pytest tests -m gpu -q -x -k "all2all_dense_bit_exact or random_forests or synthetic_databases or second_level or many_samples or (baseline_sample and 10000)" > $OUT/${TAG}_tests_var.log 2>&1; tail -3 $OUT/${TAG}_tests_var.log | cut -c1-200
timeout 900 python bench.py --mode all2all-sp --workload c4sparse > $OUT/${TAG}_c4sparse.json 2> $OUT/${TAG}_c4sparse.err; tail -2 $OUT/${TAG}_c4sparse.err | cut -c1-300
KMDB_SP_ALL_TILES=1 timeout 900 python bench.py --mode all2all-sp --workload c4sparse --no-cpu-baseline > $OUT/${TAG}_c4sparse_all.json 2> $OUT/${TAG}_c4sparse_all.err
python - <<PY
import json
for n in ("c4sparse","c4sparse_all"):
    try:
        d=json.loads(open("$OUT/${TAG}_%s.json"%n).read().strip().splitlines()[-1]); print(n, round(d["ms_per_step"],3), d["config"].get("nnz"), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("seconds"))
    except Exception as e: print(n, "no line", e)
PY
ls $OUT | grep ${TAG} | wc -l
