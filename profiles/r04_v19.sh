#!/bin/bash
# round 4, job 19 (= 18 after the page drops): default bench line; the front-end prints where its wall clock goes (process start -> main, load phases, upload parts, process up time)
OUT=$PWD/gpurun_out; mkdir -p $OUT; TAG=r04_v19
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; grep -A14 "front-end:" $OUT/${TAG}_bench.err
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json")); e=b["extra"]["c3part"]
print("c2", round(b["ms_per_step"],3), {k:v for k,v in b["wall"].items() if k.startswith("frontend") or k.startswith("reference")})
print("c3part", round(e["ms_per_step"],3), {k:v for k,v in e.items() if k.startswith("frontend") or k.startswith("reference")})
PY
