#!/bin/bash
# Round 6, job 7: the tests job 6 failed or did not have yet (glue inside the reference rebuilt against ABI 7, deep tree beyond 65 535 samples, bench.py
# --driver node, all2all-parts over workers), then the default bench line with the secondary rows riding along — and how long the whole run takes.
TAG=r06_j7
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=6 -k "integration_glue or more_than_65535 or node_driver or parts_grid or cli_byte or protein or pools_too_small" > $OUT/${TAG}_tests_sel.log 2>&1; tail -14 $OUT/${TAG}_tests_sel.log | cut -c1-220
t0=$(date +%s)
KMDB_VERBOSE=1 timeout 1500 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json
echo "default bench: $(( $(date +%s) - t0 )) s"
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json"))
print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), {k:round(v,3) for k,v in b["roofline"]["per_kernel_ms"].items()}, b["roofline"]["block_records_per_launch"], "upload", round(b["wall"]["upload_s"],3))
for n in ("c3part","c3gpu"):
    e=b["extra"][n]; print(n, round(e["ms_per_step"],3), {k:round(v,3) for k,v in e["per_kernel_ms"].items()}, e["records"], "upload", round(e.get("upload_s",0),3))
for n in ("new2all_c5part","db2db_parts"):
    e=b["extra"][n]; print(n, round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e.get("cpu_baseline",{}).get("kind"), round(e["seconds_in_bench"],1), "s")
print("cpu", b.get("cpu_baseline",{}).get("kind"), b.get("cpu_baseline",{}).get("seconds"), b.get("cpu_baseline",{}).get("sweep"))
PY
