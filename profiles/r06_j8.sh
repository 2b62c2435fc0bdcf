#!/bin/bash
# Round 6, job 8: the tests job 7 failed (deep chain that reaches ids beyond 65 535; RCCL's banner behind the node driver's line) and the new ones
# (-sample-rows against the reference's Sampler; all2all-parts over workers).
TAG=r06_j8
OUT=$PWD/gpurun_out
mkdir -p $OUT
export KMDB_REQUIRE_REF=1
timeout 1500 python -m pytest tests -m gpu -q -rs --durations=6 -k "more_than_65535 or node_driver or sample_rows or parts_grid or cli_byte or integration_glue" > $OUT/${TAG}_tests_sel.log 2>&1; tail -14 $OUT/${TAG}_tests_sel.log | cut -c1-220
