#!/bin/bash
# Round 4, job 6: K0's long launch cutting its lists from relative 64-id words kept in registers instead of walking the stream twice.
OUT=gpurun_out/r04h; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "all2all_dense_bit_exact or synthetic_databases or random_forests or randomised or many_samples or baseline_sample_counts or pools_too_small or degenerate or patterns_that_touch or shards" > $OUT/tests_sel.log 2>&1; tail -3 $OUT/tests_sel.log
WL="c3part c2" bash profiles/r04_const_ab.sh "K0_SINGLE_PASS=0" "K0_RELW=4" "K0_RELW=8" 2>&1 | tee $OUT/const_ab.txt
