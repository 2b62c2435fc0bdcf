#!/bin/bash
# Round 5, job 10: the whole GPU suite (durations; target <= 750 s) and the default bench line (C2 + extra.c3part + extra.c3gpu, reference swept at full size)
TAG=r05_j10
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/${TAG}_tests.log 2>&1; tail -40 $OUT/${TAG}_tests.log | cut -c1-200
KMDB_VERBOSE=1 timeout 1500 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; grep -v "synth build" $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.err2; mv $OUT/${TAG}_bench.err2 $OUT/${TAG}_bench.err
python - <<PY
import json
try:
    b=json.load(open("$OUT/${TAG}_bench.json"))
    print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["per_kernel_ms"], b["cpu_baseline"].get("sweep"), b["cpu_baseline"]["seconds"])
    for n in ("c3part","c3gpu"):
        e=b["extra"][n]; print(n, round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["per_kernel_ms"], e["records"], e.get("nodes_joined_per_tile"), e.get("rows_from_definition"), e.get("reference_match"))
except Exception as ex: print("bench line:", ex)
PY
grep -c . $OUT/${TAG}_bench.err
