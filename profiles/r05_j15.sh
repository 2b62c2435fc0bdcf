#!/bin/bash
# Round 5, job 15: cs_hist with eight key words per thread requested together, rs_hist at 16 chunks per round trip by default: parity of the
# few-streams / many-streams sorts, C2 and c3part once each
TAG=r05_j15
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "few_streams_record_forms or all2all_dense_bit_exact or random_forests or randomised_stress or db2db_bit_exact or second_level or pools_too_small" > $OUT/${TAG}_tests_sel.log 2>&1; tail -3 $OUT/${TAG}_tests_sel.log | cut -c1-200
for wl in "" "--workload c3part"; do
  timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 10 --warmup 3 > $OUT/${TAG}_b.json 2> $OUT/${TAG}_b.err
  python - <<PY
import json
d=json.load(open("$OUT/${TAG}_b.json")); print("$wl", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()})
PY
done
