#!/bin/bash
# Round 6, first job (prepared at the end of round 5, which ran out of GPU minutes): what round 5 changed after its last GPU job, measured.
#   1. profiles/r05_gen_check.py — the synthetic generator's short-genome forms on the GPU: equal to the loop forms and to the CPU, ms per sample
#   2. the whole GPU suite with KMDB_TEST_PHASES (round 5, job 13: 98 passed in 805 s; expected now: 106 passed in about 600 s — the
#      generator was 390 s of it)
#   3. the default bench line (replays profiles/latest_traffic*.json while the all2all sources are those of round 5's closing job)
TAG=r06_j1
OUT=$PWD/gpurun_out
mkdir -p $OUT
timeout 300 python profiles/r05_gen_check.py > $OUT/${TAG}_gen_check.txt 2>&1; tail -2 $OUT/${TAG}_gen_check.txt
KMDB_TEST_PHASES=$OUT/${TAG}_test_phases.txt timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $OUT/${TAG}_tests.log 2>&1; tail -22 $OUT/${TAG}_tests.log | cut -c1-200
cat $OUT/${TAG}_test_phases.txt
KMDB_VERBOSE=1 timeout 900 python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json"))
print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], b["roofline"]["per_kernel_ms"])
for n in ("c3part","c3gpu"):
    e=b["extra"][n]; print(n, round(e["ms_per_step"],3), e["per_kernel_ms"], e["records"])
PY
