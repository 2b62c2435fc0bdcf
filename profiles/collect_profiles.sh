#!/bin/bash
# Collect the per-round profile evidence on the GPU box (run from the repo root):
#   bash profiles/collect_profiles.sh <tag>        e.g. r01_v8
# 1. bench line (with cpu_baseline)   2. rocprofv3 kernel stats of the same command
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (never combined with other trace domains)
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${2:-all}" != "pmc" ]; then
python bench.py 2> $OUT/${TAG}_bench.err | tee $OUT/${TAG}_bench.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $OLDPWD/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $OUT/${TAG}_bench_profiled.json 2> /tmp/prof_stats.err )
find /tmp/prof_stats -name '*kernel_stats.csv' -exec cp {} $OUT/${TAG}_kernel_stats_all.csv \;
fi
for C in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "b3_|b2_" --output-format csv -d /tmp/prof_$C -- python $OLDPWD/bench.py --no-cpu-baseline --steps 2 --warmup 1 > /dev/null 2> /tmp/prof_$C.err )
  find /tmp/prof_$C -name '*counter_collection.csv' -exec cp {} /tmp/${C}.csv \;
done
python profiles/summarize_profiles.py $TAG /tmp/FETCH_SIZE.csv /tmp/WRITE_SIZE.csv $OUT
ls -la $OUT | tail -12
