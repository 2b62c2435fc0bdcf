#!/bin/bash
# Round 4, job 7: new2all walking the handle's run index instead of decoding the gamma streams again for every query.
OUT=gpurun_out/r04i; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "new2all or one2all or cli_byte or extraction or smoke" > $OUT/tests_sel.log 2>&1; tail -3 $OUT/tests_sel.log
for v in "" "KMDB_N2A_NO_RUNS=1"; do
  env $v KMDB_VERBOSE=1 timeout 900 python bench.py --mode new2all --no-cpu-baseline 2> $OUT/n2a_${v:-runs}.err > $OUT/n2a_${v:-runs}.json
  python -c "
import json; d=json.loads(open('$OUT/n2a_${v:-runs}.json').read().strip().splitlines()[-1]); print('new2all ${v:-run index}', round(d['ms_per_step'],2), d['wall'].get('call_ms'))"
  grep "run index" $OUT/n2a_${v:-runs}.err | head -2
done
