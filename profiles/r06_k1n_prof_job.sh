#!/bin/bash
OUT=$PWD/gpurun_out
for wl in c2 c3part; do
  KMDB_K1N_PROF=1 timeout 600 python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 3 --warmup 1 2> $OUT/r06_k1n_prof_$wl.err > /dev/null
  grep "narrow kernel profile" $OUT/r06_k1n_prof_$wl.err | head -1
done
