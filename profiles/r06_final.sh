#!/bin/bash
# Round 6, last job: what the driver runs at the round's end, on the final sources — the whole GPU suite, smoke(), and (quickly) the bench contract.
TAG=r06_final
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
timeout 1500 python -m pytest tests -x -q -m gpu -rs > $OUT/${TAG}_tests.log 2>&1; grep -n "passed\|failed\|skipped\|SKIP" $OUT/${TAG}_tests.log | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -2 $OUT/${TAG}_smoke.log | cut -c1-300
timeout 600 python bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline > $OUT/${TAG}_bench_quick.json 2> $OUT/${TAG}_bench_quick.err; python -c "
import json; d=json.load(open('$OUT/${TAG}_bench_quick.json')); print('c2', round(d['ms_per_step'],3), round(d['roofline']['frac'],4), d['roofline']['traffic'])"
