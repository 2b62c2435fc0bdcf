#!/bin/bash
# (1) the test beyond 65 535 samples; (2) same-box A/B of the wide kernel: many-block nodes emitted row by row (K1W_HEAVY_ROWS), 16 instead of
# 14 waves per CU (smaller queue / own-pair cache), the threshold of "many blocks"; (3) the parity tests that lean on the wide kernel
OUT=$PWD/gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x -k "more_than_65535" > $OUT/r04_v11_tests_big.log 2>&1; tail -5 $OUT/r04_v11_tests_big.log
WL="c3part c2" bash profiles/r04_const_ab.sh "K1W_HEAVY_ROWS=0" "K1W_QCAP=64,K1W_OXCAP=64" "K1W_QCAP=64,K1W_OXCAP=64,K1W_HEAVY=8" "K1W_QCAP=64,K1W_OXCAP=64,K1W_HEAVY=16" "K1W_HEAVY_ROWS=0,K1W_QCAP=64,K1W_OXCAP=64" > $OUT/r04_v11_const_ab_k1w_rows.txt 2>&1
cat $OUT/r04_v11_const_ab_k1w_rows.txt
timeout 1500 python -m pytest tests -m gpu -q -x -k "many_samples or baseline_sample_counts or random_forests or patterns_that_touch or pools_too_small or synthetic_databases or randomised_stress" > $OUT/r04_v11_tests_wide.log 2>&1; tail -5 $OUT/r04_v11_tests_wide.log
