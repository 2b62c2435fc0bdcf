#!/bin/bash
# Round 4, job 8: runs per piece of a queued long list in the new2all walk (compile-time), same box.
OUT=gpurun_out/r04j; mkdir -p $OUT
cp kmer-db_amd/csrc/new2all.hip /tmp/new2all.orig
for v in 8 4 16 32; do
  sed -E "s/N2_RUNS_PER_PIECE = [0-9]+/N2_RUNS_PER_PIECE = $v/" /tmp/new2all.orig > kmer-db_amd/csrc/new2all.hip
  make -C kmer-db_amd -j8 > /dev/null 2>&1
  timeout 600 python bench.py --mode new2all --no-cpu-baseline 2> /dev/null > $OUT/n2a_rpp$v.json
  python -c "
import json; d=json.loads(open('$OUT/n2a_rpp$v.json').read().strip().splitlines()[-1]); print('runs per piece $v:', round(d['ms_per_step'],2))"
done
cp /tmp/new2all.orig kmer-db_amd/csrc/new2all.hip; make -C kmer-db_amd -j8 > /dev/null 2>&1
