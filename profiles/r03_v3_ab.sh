#!/bin/bash
# db2db with packed keys; new2all walk with 1024 / 512 threads per workgroup; kernel stats of c3part and of the two secondary modes
OUT=$PWD/gpurun_out; mkdir -p $OUT
python -m pytest tests/test_gpu_parity.py -q -x -k "db2db or new2all or one2all or parts" 2>&1 | tail -3
for t in 1024 512; do
  KMDB_N2A_THREADS=$t python bench.py --mode new2all --no-cpu-baseline 2> /dev/null > $OUT/ab_n2a_$t.json
  python -c "
import json; d=json.loads(open('$OUT/ab_n2a_$t.json').read().strip().splitlines()[-1]); print('new2all threads $t', d['ms_per_step'], d['wall']['call_ms'])"
done
python bench.py --mode db2db 2> $OUT/r03_v3_mode_db2db.err > $OUT/r03_v3_mode_db2db.json
python bench.py --mode new2all 2> $OUT/r03_v3_mode_new2all.err > $OUT/r03_v3_mode_new2all.json
python -c "
import json
for m in ('db2db','new2all'):
    d=json.loads(open('$OUT/r03_v3_mode_%s.json'%m).read().strip().splitlines()[-1]); print(m, d['ms_per_step'], d['wall'], d.get('cpu_baseline',{}).get('sample'))"
BENCH_ARGS="--mode db2db" bash profiles/collect_profiles.sh r03_v3_mode_db2db stats > $OUT/r03_v3_cp_db2db.log 2>&1
BENCH_ARGS="--mode new2all" bash profiles/collect_profiles.sh r03_v3_mode_new2all stats > $OUT/r03_v3_cp_new2all.log 2>&1
BENCH_ARGS="--workload c3part" bash profiles/collect_profiles.sh r03_v2_c3part stats > $OUT/r03_v2_cp_c3.log 2>&1
rm -f $OUT/*_kernel_stats_all.csv
head -12 $OUT/r03_v3_mode_db2db_kernel_stats.csv | cut -c1-100
head -6 $OUT/r03_v3_mode_new2all_kernel_stats.csv | cut -c1-100
head -8 $OUT/r03_v2_c3part_kernel_stats.csv | cut -c1-100
