#!/bin/bash
# Same-box A/B of run-time knobs (environment variables read by the engine): bench.py once per setting, whole-call ms and the stages.
#   BENCH_ARGS="--workload c3part" bash profiles/r03_env_ab.sh "KMDB_K1W_WAVES=2048" "KMDB_K1W_WAVES=1024 KMDB_K1W_RUN=512"
BA=${BENCH_ARGS:-}
for V in "" "$@"; do
  echo "== ${V:-default}"
  env $V python bench.py $BA --no-cpu-baseline --no-extra --steps 5 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_step'],3), {k: round(v,2) for k,v in d['roofline']['per_kernel_ms'].items()}, 'records', d['roofline']['block_records_per_launch'])"
done
