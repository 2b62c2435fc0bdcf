#!/bin/bash
# Round-3 final pass on the GPU box: the whole GPU test suite, the default bench line, the secondary modes' lines and kernel stats
TAG=${1:-r03_v3}
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/${TAG}_tests.log 2>&1; tail -3 $OUT/${TAG}_tests.log
python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; tail -c 300 $OUT/${TAG}_bench.json
for m in db2db new2all; do
  python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json
  BENCH_ARGS="--mode $m" bash profiles/collect_profiles.sh ${TAG}_mode_$m stats > $OUT/${TAG}_cp_$m.log 2>&1
done
rm -f $OUT/*_kernel_stats_all.csv
python - <<PY
import json
for m in ("db2db","new2all"):
    d=json.loads(open("$OUT/${TAG}_mode_%s.json"%m).read().strip().splitlines()[-1]); print(m, d["ms_per_step"], d["wall"])
PY
