#!/bin/bash
# new2all: the climb reads one 16-byte record per node (built with the run index) instead of five loads from four arrays (KMDB_N2A_NO_NODES=1), same box
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "new2all or more_than_65535 or extraction" > $OUT/r04_v16_tests_n2a.log 2>&1; tail -3 $OUT/r04_v16_tests_n2a.log
python bench.py --mode new2all --no-cpu-baseline 2> $OUT/r04_v16_mode_new2all_nodes.err > $OUT/r04_v16_mode_new2all_nodes.json
KMDB_N2A_NO_NODES=1 python bench.py --mode new2all --no-cpu-baseline 2> /dev/null > $OUT/r04_v16_mode_new2all_nonodes.json
python bench.py --mode new2all --no-cpu-baseline 2> /dev/null > $OUT/r04_v16_mode_new2all_nodes_again.json
for f in nodes nonodes nodes_again; do python - <<PY
import json
d=json.loads(open("$OUT/r04_v16_mode_new2all_$f.json").read().strip().splitlines()[-1]); print("$f", round(d["ms_per_step"],3), d["roofline"]["frac"], d["roofline"]["union_of_root_paths"]["frac"])
PY
done
