#!/bin/bash
# second job with the patched library (see r04_l2_try.sh): the threshold at the wide kernel's own (11 blocks) — more nodes on the second level —
# under the randomised stress test and the random forests (row mode forced in part of their cases), then c3part again
OUT=$PWD/gpurun_out; mkdir -p $OUT
cp kmer-db_amd/libkmdb_amd.so /tmp/libkmdb_amd_committed.so
cp profiles/_l2/libkmdb_amd_l2.so kmer-db_amd/libkmdb_amd.so
KMDB_L2_MIN=11 timeout 100 python -m pytest tests -m gpu -q -x -k "randomised_stress or random_forests" > $OUT/r04_l2_test2.log 2>&1; tail -3 $OUT/r04_l2_test2.log
KMDB_L2_MIN=11 KMDB_VERBOSE=1 timeout 80 python bench.py --workload c3part --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r04_l2_c3part_min11.json 2> $OUT/r04_l2_c3part_min11.err; grep -E "second level|doubling|Assertion|Error" $OUT/r04_l2_c3part_min11.err | head -5
python - <<PY
import json
try:
    d=json.load(open("$OUT/r04_l2_c3part_min11.json")); print("c3part, threshold 11:", round(d["ms_per_step"],3), d["roofline"]["per_kernel_ms"], d["roofline"]["block_records_per_launch"])
except Exception as e: print("no bench line:", e)
PY
cp /tmp/libkmdb_amd_committed.so kmer-db_amd/libkmdb_amd.so
