#!/bin/bash
# Round 6, job 14: the narrow kernel's summary merge (nsum_merge: run 3.3 - 3.7 times per batch by the pointer doubling) branch-free — selects on the
# fields instead of six cases as divergent branches.  Same-job A/B by swapping the library (_ab_old.so: the committed form, _ab_new.so), parity first.
TAG=r06_j14
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
cp _ab_new.so kmer-db_amd/libkmdb_amd.so
timeout 1200 python -m pytest tests -m gpu -q -x -k "all2all or synthetic_databases or random_forests or randomised_stress or pools_too_small or degenerate or second_level or many_samples or touch_every_block or 10000-50-400 or few_streams or upload_shards or sparse or more_than_65535" > $OUT/${TAG}_tests_sel.log 2>&1; tail -3 $OUT/${TAG}_tests_sel.log | cut -c1-200
ab() {  # name, lib, workload args
  name=$1; lib=$2; wl=$3
  cp $lib kmer-db_amd/libkmdb_amd.so
  timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()})
except Exception as e: print("$name: no line", e)
PY
}
ab c2_old_a _ab_old.so ""
ab c2_new_a _ab_new.so ""
ab c2_old_b _ab_old.so ""
ab c2_new_b _ab_new.so ""
ab c3_old _ab_old.so "--workload c3part"
ab c3_new _ab_new.so "--workload c3part"
ab c3gpu_old _ab_old.so "--workload c3gpu"
ab c3gpu_new _ab_new.so "--workload c3gpu"
cp _ab_new.so kmer-db_amd/libkmdb_amd.so
