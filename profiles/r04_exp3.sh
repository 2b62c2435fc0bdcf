#!/bin/bash
# Round 4, experiment job 3: the apply kernels without LDS round trips for the operand rows (v_permlane32_swap), VALU-only transposes, ALU
# byte spreading; LDS footprint of the wide kernel; sort tiles.
OUT=gpurun_out/r04e; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -k "all2all_dense_bit_exact or synthetic_databases or random_forests or randomised or db2db_bit_exact or degenerate or patterns_that_touch" > $OUT/tests_sel.log 2>&1; tail -3 $OUT/tests_sel.log
WL="c3part c2" bash profiles/r04_const_ab.sh "K2_TR_SWIZZLE=1" "K2_SPREAD_LUT=0" "K2_SPREAD_LUT=0,K2_TR_SWIZZLE=1" "K1W_ARENA_MIN=256" "K1W_ARENA_MIN=256,K1W_OXCAP=128" "K1W_ARENA_MIN=384" "CS_TILE=2048" 2>&1 | tee $OUT/const_ab.txt
