#!/bin/bash
# wide kernel: two steps of 64 records in flight (pairs computed side by side before their stores) against one; the committed variant runs first and
# last (the sort + apply of later processes of a job has been seen 0.8 ms slower than the first one's: r04_v11 / r04_v12)
OUT=$PWD/gpurun_out; mkdir -p $OUT
WL="c3part c2" bash profiles/r04_const_ab.sh "K1W_STEPS=1" "K1W_STEPS=2" "K1W_STEPS=1" > $OUT/r04_v13_const_ab_k1w_steps.txt 2>&1
cat $OUT/r04_v13_const_ab_k1w_steps.txt
timeout 1200 python -m pytest tests -m gpu -q -x -k "many_samples or random_forests or patterns_that_touch or pools_too_small or synthetic_databases" > $OUT/r04_v13_tests_wide.log 2>&1; tail -3 $OUT/r04_v13_tests_wide.log
