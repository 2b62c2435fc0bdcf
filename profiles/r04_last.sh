#!/bin/bash
# last job of round 4: what the driver runs at the end of the round, on the last commit — smoke() and the default bench line with the driver's flags
OUT=$PWD/gpurun_out; mkdir -p $OUT; TAG=r04_last
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -3 $OUT/${TAG}_smoke.log
python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; grep -A14 "^front-end:" $OUT/${TAG}_bench.err | head -34
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json")); e=b["extra"]["c3part"]
print("c2", round(b["ms_per_step"],3), round(b["roofline"]["frac"],4), b["roofline"]["traffic"], {k:round(v,3) for k,v in b["wall"].items() if k.startswith("frontend") and isinstance(v,float)})
print("c3part", round(e["ms_per_step"],3), round(e["roofline"]["frac"],4), e["roofline"]["traffic"], {k:round(v,3) for k,v in e.items() if k.startswith("frontend") and isinstance(v,float)})
PY
