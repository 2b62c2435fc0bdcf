#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run from the repo root):
#   bash profiles/collect_profiles.sh <tag> [all|stats|pmc]        e.g. r02_v3        (BENCH_ARGS="--workload c3part" for another workload)
# 1. bench line (with cpu_baseline)   2. rocprofv3 kernel stats of the same command
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (never combined with other trace domains)
set -u
TAG=${1:-rXX}
WHAT=${2:-all}
OUT=$PWD/gpurun_out
R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
BA=${BENCH_ARGS:-}
if [ "$WHAT" = "all" ]; then
python bench.py $BA 2> $OUT/${TAG}_bench.err | tee $OUT/${TAG}_bench.json
fi
if [ "$WHAT" != "pmc" ]; then
rm -rf /tmp/prof_stats
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py $BA --no-cpu-baseline --steps 5 --warmup 2 > $OUT/${TAG}_bench_profiled.json 2> /tmp/prof_stats.err )
# one stats file per traced process (bench.py runs its synthetic generator in a child): keep the one with the engine's kernels
for f in $(find /tmp/prof_stats -name '*kernel_stats.csv'); do
  if grep -q -E "k0_decode_kernel|n2a_|d2_|row_nnz_kernel" $f; then cp $f $OUT/${TAG}_kernel_stats_all.csv; fi
done
fi
if [ "$WHAT" != "stats" ]; then
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_$C
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/prof_$C -- python $R/bench.py $BA --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> /tmp/prof_$C.err )
  cp $(find /tmp/prof_$C -name '*counter_collection.csv' -printf '%s %p\n' | sort -n | tail -1 | cut -d' ' -f2) /tmp/${C}.csv
done
fi
python profiles/summarize_profiles.py $TAG /tmp/FETCH_SIZE.csv /tmp/WRITE_SIZE.csv $OUT
ls -la $OUT | tail -12
