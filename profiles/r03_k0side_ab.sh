#!/bin/bash
# A/B: K0 side streams on/off at c2 and c3part; db2db with / without the list store
cd $GRAFT_REPO_ROOT
for side in 1 0; do
  for wl in c2 c3part; do
    KMDB_K0_SIDE=$side python bench.py --workload $wl --no-cpu-baseline --no-extra --steps 8 --warmup 3 2> gpurun_out/ab_k0side_${side}_$wl.err > gpurun_out/ab_k0side_${side}_$wl.json
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab_k0side_${side}_$wl.json").read().strip().splitlines()[-1])
print("K0_SIDE=$side $wl", round(d["ms_per_step"],3), {k: round(v,2) for k,v in d["roofline"]["per_kernel_ms"].items()})
PY
  done
done
python -m pytest tests/test_gpu_parity.py -q -x -k "db2db" 2>&1 | tail -3
KMDB_VERBOSE=1 python bench.py --mode db2db --no-cpu-baseline 2> gpurun_out/ab_db2db_store.err > gpurun_out/ab_db2db_store.json; python -c "
import json; d=json.loads(open('gpurun_out/ab_db2db_store.json').read().strip().splitlines()[-1]); print('db2db store', d['ms_per_step'], d['wall'])"
grep "list store" gpurun_out/ab_db2db_store.err | head -3
KMDB_D2_NO_STORE=1 python bench.py --mode db2db --no-cpu-baseline 2> /dev/null > gpurun_out/ab_db2db_nostore.json; python -c "
import json; d=json.loads(open('gpurun_out/ab_db2db_nostore.json').read().strip().splitlines()[-1]); print('db2db no store', d['ms_per_step'], d['wall'])"
grep "prepare:" gpurun_out/ab_k0side_1_c3part.err
