#!/bin/bash
# round 4, job 17: the RCCL-on-one-rank tests, the front-end tests, and the default bench line with the front-end's new timings
# (parallel .db reader, device prepared next to the read, process ended at once; the same command with the full teardown beside it)
OUT=$PWD/gpurun_out; mkdir -p $OUT; TAG=r04_v17
python -m pytest tests -m gpu -q -k "one_rank or dense_bit_exact or cli_byte or node_driver or integration_glue" > $OUT/${TAG}_tests_sel.log 2>&1; tail -15 $OUT/${TAG}_tests_sel.log
KMDB_VERBOSE=1 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; grep -v "synth build" $OUT/${TAG}_bench.err | tail -5
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json")); e=b["extra"]["c3part"]
print("c2", round(b["ms_per_step"],3), {k:v for k,v in b["wall"].items() if k.startswith("frontend") or k.startswith("reference")})
print("c3part", round(e["ms_per_step"],3), {k:v for k,v in e.items() if k.startswith("frontend") or k.startswith("reference")})
PY
