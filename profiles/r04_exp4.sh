#!/bin/bash
# Round 4, experiment job 4: XCD-contiguous job order in the sort kernels, 2048-record sort tiles, smaller LDS footprint of the wide kernel;
# where the sorted-records apply kernel spends its time (parts switched off at compile time: timing only).
OUT=gpurun_out/r04f; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "all2all_dense_bit_exact or synthetic_databases or random_forests or randomised or db2db or degenerate or patterns_that_touch or many_samples or pools_too_small or baseline_sample_counts" > $OUT/tests_sel.log 2>&1; tail -3 $OUT/tests_sel.log
WL="c3part c2" bash profiles/r04_const_ab.sh "RS_XCD_MAP=0" "CS_TILE=1024" "K1W_ARENA_MIN=512,K1W_OXCAP=256" 2>&1 | tee $OUT/const_ab.txt
WL="c3part" bash profiles/r04_const_ab.sh "K2_DEBUG=1" "K2_DEBUG=2" "K2_DEBUG=3" "K2_DEBUG=4" "K2_DEBUG=5" 2>&1 | tee $OUT/k2_parts.txt
