#!/bin/bash
# Round 5, closing job, second part (what the first part's clock left out): the secondary modes' bench lines on the final sources
TAG=r05_close
OUT=$PWD/gpurun_out
mkdir -p $OUT
for m in new2all db2db all2all-sp; do
  [ $SECONDS -lt 230 ] && timeout 110 python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json
  echo "$m done at ${SECONDS}s"
done
python - <<PY
import json
for m in ("all2all-sp","new2all","db2db"):
    try:
        d=json.loads(open("$OUT/${TAG}_mode_%s.json"%m).read().strip().splitlines()[-1]); print(m, round(d["ms_per_step"],3), d["roofline"]["frac"], (d.get("cpu_baseline") or {}).get("seconds"))
    except Exception as ex: print(m, "no line", ex)
PY
