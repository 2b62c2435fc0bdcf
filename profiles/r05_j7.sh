#!/bin/bash
# Round 5, job 7: what the wide kernel's batches look like (profiling build: rounds per batch, entries of lists and rows, walk steps)
TAG=r05_j7
OUT=$PWD/gpurun_out
mkdir -p $OUT
for wl in "--workload c3part" ""; do
  n=$(echo $wl | tr -d ' -' ); n=${n:-c2}
  KMDB_K1W_PROF=1 timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 > $OUT/${TAG}_$n.json 2> $OUT/${TAG}_$n.err
  grep -h "k1w phases\|k1w counts" $OUT/${TAG}_$n.err | tail -2
done
