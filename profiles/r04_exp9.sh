#!/bin/bash
# 20-bit sample ids: the new test beyond 65 535 samples, the parity tests closest to the changed packings, and the C2 / 10 000-sample timing on the new layout
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "more_than_65535 or random_forests or degenerate or new2all_bit_exact or all2all_dense_bit_exact or new2all_synthetic_scale" > $OUT/r04_v10_tests.log 2>&1; tail -15 $OUT/r04_v10_tests.log
python bench.py --no-cpu-baseline 2> $OUT/r04_v10_bench.err > $OUT/r04_v10_bench.json; tail -3 $OUT/r04_v10_bench.err
python - <<PY
import json
b=json.load(open("$OUT/r04_v10_bench.json")); e=b["extra"]["c3part"]
print("c2", round(b["ms_per_step"],3), b["roofline"]["per_kernel_ms"]); print("c3part", round(e["ms_per_step"],3), e["per_kernel_ms"])
PY
