#!/bin/bash
# The second level above the block records, integrated (profiles/r04_level2_integration.diff applied in a scratch tree, built there, the library
# shipped as profiles/_l2/libkmdb_amd_l2.so): tried on the GPU box WITHOUT touching the committed sources — the box's copy of the library is
# swapped for the patched one.  1. the 10 000-sample parity test (whole matrix == oracle)  2. c3part warm calls (checksum identity, warm == cold)
OUT=$PWD/gpurun_out; mkdir -p $OUT
cp kmer-db_amd/libkmdb_amd.so /tmp/libkmdb_amd_committed.so
cp profiles/_l2/libkmdb_amd_l2.so kmer-db_amd/libkmdb_amd.so
KMDB_VERBOSE=1 timeout 170 python -m pytest tests -m gpu -q -x -k "baseline_sample_counts and 10000" -s > $OUT/r04_l2_test.log 2>&1; grep -E "second level|passed|failed|Error|error" $OUT/r04_l2_test.log | head -8
KMDB_VERBOSE=1 timeout 150 python bench.py --workload c3part --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r04_l2_c3part.json 2> $OUT/r04_l2_c3part.err; grep -E "second level|doubling|Assertion|Error" $OUT/r04_l2_c3part.err | head -5
python - <<PY
import json
try:
    d=json.load(open("$OUT/r04_l2_c3part.json")); print("c3part with the second level:", round(d["ms_per_step"],3), d["roofline"]["per_kernel_ms"], d["roofline"]["block_records_per_launch"])
except Exception as e: print("no bench line:", e)
PY
cp /tmp/libkmdb_amd_committed.so kmer-db_amd/libkmdb_amd.so
