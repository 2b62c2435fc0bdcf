#!/bin/bash
# Round 4, job 5: four-slot hash probing (new2all / db2db tests + bench lines with the reference on all useful threads), per-kernel times
# and SQ counters of the all2all call on the current code.
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "new2all or db2db or cli_byte or one2all or extraction" > $OUT/r04_v6_tests_sel.log 2>&1; tail -3 $OUT/r04_v6_tests_sel.log
BENCH_ARGS="--workload c3part --no-extra" bash profiles/collect_profiles.sh r04_v6_c3part stats > $OUT/r04_v6_cp_c3.log 2>&1
BENCH_ARGS="--no-extra" bash profiles/collect_profiles.sh r04_v6_c2 stats > $OUT/r04_v6_cp_c2.log 2>&1
BENCH_ARGS="--workload c3part --no-extra" bash profiles/collect_counters.sh r04_v6_c3part sq1 sq2 sq3 > $OUT/r04_v6_cc_c3.log 2>&1
for m in new2all db2db; do
  timeout 900 python bench.py --mode $m 2> $OUT/r04_v6_mode_$m.err > $OUT/r04_v6_mode_$m.json
  python -c "
import json; d=json.loads(open('$OUT/r04_v6_mode_$m.json').read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('union_of_root_paths'), d.get('cpu_baseline'))"
done
rm -f $OUT/*_kernel_stats_all.csv
python - <<'PY'
import csv
for w in ("c3part","c2"):
    try:
        rows=list(csv.DictReader(open("gpurun_out/r04_v6_%s_kernel_stats.csv"%w)))
        print(w, [(r.get("Name","")[:28], r.get("AverageNs") or r.get("Average")) for r in rows[:14]])
    except Exception as e: print(w, "no stats", e)
PY
