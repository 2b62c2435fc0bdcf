#!/bin/bash
# Round-3 evidence in one go on the GPU box (run from the repo root):  bash profiles/r03_collect_all.sh <tag>
#   1. the default bench line (c2 + cpu_baseline + the 10 000-sample workload riding along, both compared with the real reference)
#   2. rocprofv3 --kernel-trace --stats of c2 and c3part
#   3. per-kernel hardware counters (SQ groups, FETCH_SIZE, WRITE_SIZE in separate --pmc passes) of c2 and c3part
#   4. the secondary modes (all2all-sp, new2all, db2db): bench lines, kernel stats, counters of their kernels
TAG=${1:-r03}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python bench.py 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json; tail -2 $OUT/${TAG}_bench.err
BENCH_ARGS="--no-extra" bash profiles/collect_profiles.sh ${TAG}_c2 stats > $OUT/${TAG}_cp_c2.log 2>&1
BENCH_ARGS="--workload c3part" bash profiles/collect_profiles.sh ${TAG}_c3part stats > $OUT/${TAG}_cp_c3.log 2>&1
BENCH_ARGS="--no-extra" bash profiles/collect_counters.sh ${TAG}_c2 sq1 sq2 sq3 fetch write > $OUT/${TAG}_cc_c2.log 2>&1
BENCH_ARGS="--workload c3part" bash profiles/collect_counters.sh ${TAG}_c3part sq1 sq2 sq3 fetch write > $OUT/${TAG}_cc_c3.log 2>&1
for m in all2all-sp new2all db2db; do
  python bench.py --mode $m 2> $OUT/${TAG}_mode_$m.err > $OUT/${TAG}_mode_$m.json
  BENCH_ARGS="--mode $m" bash profiles/collect_profiles.sh ${TAG}_mode_$m stats > $OUT/${TAG}_cp_$m.log 2>&1
  BENCH_ARGS="--mode $m" bash profiles/collect_counters.sh ${TAG}_mode_$m sq1 sq2 > $OUT/${TAG}_cc_$m.log 2>&1
done
ls -la $OUT | grep ${TAG} | head -60
