#!/bin/bash
# Round 6, job 12: the many-streams record pool sized for the chunks the waves leave partly filled (the closing job's first calls at 10 000 samples ran
# out of chunks at 1.25 x the estimate and doubled the pool): chunks in use against chunks allocated, no enlarge-and-repeat, upload and cold call.
TAG=r06_j12
OUT=$PWD/gpurun_out
mkdir -p $OUT
for wl in c3part c3gpu; do
  KMDB_VERBOSE=1 timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 5 --warmup 2 > $OUT/${TAG}_$wl.json 2> $OUT/${TAG}_$wl.err
  grep -n "record pool\|too small\|enlarged\|estimated\|cold call\|upload: total\|hipMalloc" $OUT/${TAG}_$wl.err | cut -c1-200
  python - <<PY
import json
d=json.load(open("$OUT/${TAG}_$wl.json")); print("$wl", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, "upload", round(d["wall"]["upload_s"],3), "cold", round(d["wall"]["cold_call_ms"],1))
PY
done
timeout 600 python -m pytest tests -m gpu -q -k "pools_too_small or 10000-50-400 or second_level or many_samples" > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log | cut -c1-200
