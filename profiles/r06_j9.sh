#!/bin/bash
# Round 6, job 9: gamma codes of more than 32 bits in the decode launches (deltas of 2^16 and more, collections beyond 65 536 samples): the
# forest of test_more_than_65535_samples now holds lists that jump over 65 536 ids at once; the decode-heavy parity tests; C2 once (decode ms).
TAG=r06_j9
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=5 -k "more_than_65535 or random_forests or all2all_dense_bit_exact or randomised_stress or degenerate or node_driver_over" > $OUT/${TAG}_tests_sel.log 2>&1; tail -10 $OUT/${TAG}_tests_sel.log | cut -c1-220
timeout 400 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_c2.json 2> $OUT/${TAG}_c2.err
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_c2.json")); print("c2", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()})
PY
