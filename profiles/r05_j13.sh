#!/bin/bash
# Round 5, job 13: the whole GPU suite on the sources with the packed few-streams path (durations; target <= 750 s), smoke
TAG=r05_j13
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log | cut -c1-200
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $OUT/${TAG}_tests.log 2>&1; tail -40 $OUT/${TAG}_tests.log | cut -c1-200
