#!/bin/bash
# Round 4, experiment job 2: the narrow kernel decoding the short streams itself (parity tests first), deeper record prefetch in the apply kernels.
OUT=gpurun_out/r04d; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q -k "all2all_dense_bit_exact or synthetic_databases or random_forests or randomised or many_samples or baseline_sample_counts or pools_too_small or degenerate or patterns_that_touch or shards or cli_byte" > $OUT/tests_sel.log 2>&1; tail -3 $OUT/tests_sel.log
WL="c3part c2" bash profiles/r04_const_ab.sh "@KMDB_K1N_FUSED=0" "K2_PF=1" "K2_PF=1@KMDB_K1N_FUSED=0" "K2_PF=3,K2S_MIN_WAVES=2,K2A_MIN_WAVES=2" "K2_PF=4,K2S_MIN_WAVES=2,K2A_MIN_WAVES=2" "K2_PF=3" 2>&1 | tee $OUT/const_ab.txt
