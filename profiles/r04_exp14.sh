#!/bin/bash
# new2all: hits as 32-bit keys sorted inside groups of 128 queries against the device-wide sort of 64-bit keys (KMDB_N2A_SORT64=1), same box; parity first
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "new2all or more_than_65535 or extraction" > $OUT/r04_v15_tests_n2a.log 2>&1; tail -3 $OUT/r04_v15_tests_n2a.log
python bench.py --mode new2all --no-cpu-baseline 2> $OUT/r04_v15_mode_new2all_sort32.err > $OUT/r04_v15_mode_new2all_sort32.json
KMDB_N2A_SORT64=1 python bench.py --mode new2all --no-cpu-baseline 2> $OUT/r04_v15_mode_new2all_sort64.err > $OUT/r04_v15_mode_new2all_sort64.json
python bench.py --mode new2all --no-cpu-baseline 2> /dev/null > $OUT/r04_v15_mode_new2all_sort32_again.json
for f in sort32 sort64 sort32_again; do python - <<PY
import json
d=json.loads(open("$OUT/r04_v15_mode_new2all_$f.json").read().strip().splitlines()[-1]); print("$f", round(d["ms_per_step"],3), d["wall"])
PY
done
