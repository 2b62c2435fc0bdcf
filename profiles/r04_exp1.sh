#!/bin/bash
# Round 4, experiment job 1 (one GPU box): the new bit transposes checked and timed, K1w's time split by switching parts of it off
# (timing only), the default bench line on the new K2, and the tests that touch what changed.
OUT=gpurun_out/r04c; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/r04_transpose_probe profiles/r04_transpose_probe.hip 2>/dev/null && /tmp/r04_transpose_probe | tee $OUT/transpose_probe.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "integration_glue or randomised or extraction_fuzz or cli_byte or node_driver or all2all_dense_bit_exact or bench_contract" > $OUT/tests_sel.log 2>&1; tail -3 $OUT/tests_sel.log
timeout 600 python profiles/r04_ab.py c3part "" "KMDB_K1W_DEBUG=1" "KMDB_K1W_DEBUG=2" "KMDB_K1W_DEBUG=3" 2> $OUT/ab_c3part.err | tee $OUT/ab_c3part.jsonl
timeout 600 python profiles/r04_ab.py c2 "" "KMDB_K1W_DEBUG=1" "KMDB_K1W_DEBUG=2" 2> $OUT/ab_c2.err | tee $OUT/ab_c2.jsonl
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo bench rc=$?
python - <<'PY'
import json
b=json.load(open('gpurun_out/r04c/bench.json'))
print('c2', round(b['ms_per_step'],3), b['roofline']['per_kernel_ms'])
e=b['extra']['c3part']; print('c3part', round(e['ms_per_step'],3), e['per_kernel_ms'])
PY
