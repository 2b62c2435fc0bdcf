#!/bin/bash
# third job with the patched library: node indices handed out interleaved (dense from 0 on) and the tile jobs scanning only the bitmap words in use
OUT=$PWD/gpurun_out; mkdir -p $OUT
cp kmer-db_amd/libkmdb_amd.so /tmp/libkmdb_amd_committed.so
cp profiles/_l2/libkmdb_amd_l2.so kmer-db_amd/libkmdb_amd.so
KMDB_VERBOSE=1 timeout 60 python bench.py --workload c3part --no-cpu-baseline --steps 5 --warmup 2 > $OUT/r04_l2_c3part_v2.json 2> $OUT/r04_l2_c3part_v2.err; grep -E "second level|doubling|Assertion|Error" $OUT/r04_l2_c3part_v2.err | head -5
python - <<PY
import json
try:
    d=json.load(open("$OUT/r04_l2_c3part_v2.json")); print("c3part, dense indices:", round(d["ms_per_step"],3), d["roofline"]["per_kernel_ms"], d["roofline"]["block_records_per_launch"])
except Exception as e: print("no bench line:", e)
PY
cp /tmp/libkmdb_amd_committed.so kmer-db_amd/libkmdb_amd.so
