#!/bin/bash
# Round-3 closing evidence on the GPU box (run from the repo root):  bash profiles/r03_collect_v2.sh <tag>
#   1. HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) of c2 and c3part on THIS code, copied to profiles/latest_traffic*.json
#      of the box's copy so that the bench line that follows replays numbers measured on the same sources
#   2. the default bench line   3. rocprofv3 --kernel-trace --stats of c2 and c3part   4. secondary modes: bench lines + kernel stats
TAG=${1:-r03_v2}
OUT=$PWD/gpurun_out
mkdir -p $OUT
BENCH_ARGS="--no-extra" bash profiles/collect_counters.sh ${TAG}_c2 fetch write > $OUT/${TAG}_cc_c2.log 2>&1
BENCH_ARGS="--workload c3part" bash profiles/collect_counters.sh ${TAG}_c3part fetch write > $OUT/${TAG}_cc_c3.log 2>&1
[ -s $OUT/${TAG}_c2_traffic.json ] && cp $OUT/${TAG}_c2_traffic.json profiles/latest_traffic.json
[ -s $OUT/${TAG}_c3part_traffic.json ] && cp $OUT/${TAG}_c3part_traffic.json profiles/latest_traffic_c3part.json
python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; tail -2 $OUT/${TAG}_bench.err
BENCH_ARGS="--no-extra" bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
BENCH_ARGS="--workload c3part" bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
for m in db2db new2all all2all-sp; do
  python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json
  BENCH_ARGS="--mode $m" bash profiles/collect_profiles.sh ${TAG}_mode_$m stats > $OUT/${TAG}_cp_$m.log 2>&1
done
rm -f $OUT/*_kernel_stats_all.csv
ls -la $OUT | grep ${TAG} | head -60
