python -m pytest tests/test_gpu_parity.py -q -x -k "new2all or one2all" 2>&1 | tail -2
for q in 1024 256; do
  KMDB_N2A_QCAP=$q python bench.py --mode new2all --no-cpu-baseline 2> /dev/null > gpurun_out/ab_n2a_q$q.json
  python -c "
import json; d=json.loads(open('gpurun_out/ab_n2a_q$q.json').read().strip().splitlines()[-1]); print('new2all qcap $q', d['ms_per_step'], d['wall']['call_ms'])"
done
for i in 1 2; do python bench.py --mode db2db --no-cpu-baseline 2> /dev/null > gpurun_out/d2n.json; python -c "
import json; d=json.loads(open('gpurun_out/d2n.json').read().strip().splitlines()[-1]); print('db2db', d['ms_per_step'], d['wall']['call_ms'])"; done
