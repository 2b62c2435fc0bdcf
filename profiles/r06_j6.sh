#!/bin/bash
# Round 6, job 6: the whole GPU suite on the round's state so far (first-block records compacted per slice + k2d_kernel; protein alphabets in the
# query loader; HBM-atomics fallback beyond 65 535 samples; bench.py --driver node; pools sized after the second level), KMDB_REQUIRE_REF=1,
# then the default bench line (upload_s / records / pools of c3part and c3gpu in its stderr).
TAG=r06_j6
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
KMDB_TEST_PHASES=$OUT/${TAG}_test_phases.txt timeout 2400 python -m pytest tests -m gpu -q -rs --durations=12 > $OUT/${TAG}_tests.log 2>&1; tail -25 $OUT/${TAG}_tests.log | cut -c1-220
KMDB_VERBOSE=1 timeout 1200 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json"))
print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["per_kernel_ms"], b["roofline"]["block_records_per_launch"], b["roofline"]["first_block_records_per_launch"], "upload", round(b["wall"]["upload_s"],3))
for n in ("c3part","c3gpu"):
    e=b["extra"][n]; print(n, round(e["ms_per_step"],3), e["per_kernel_ms"], e["records"], e.get("records_applied_from_slices"), "upload", e.get("upload_s"))
print("cpu", b.get("cpu_baseline",{}).get("kind"), b.get("cpu_baseline",{}).get("seconds"))
PY
grep -n "record pools\|second level\|estimated" $OUT/${TAG}_bench.err | head -20
