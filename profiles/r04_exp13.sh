#!/bin/bash
# timing experiment: a second K0 (all nodes, into scratch buffers) on a side stream next to the wide kernel — does K0's issue-bound work hide under
# K1w's waiting?  whole call with it minus whole call without = what of K0's 3.8 ms (10 000 samples) / 2.3 ms (C2) does NOT hide
OUT=$PWD/gpurun_out; mkdir -p $OUT
for w in c3part c2; do python profiles/r04_ab.py $w "" "KMDB_EXP_K0_BESIDE_K1W=1" "KMDB_EXP_K0_BESIDE_K1W=2" "" 2>/dev/null; done > $OUT/r04_v14b_ab_k0_beside_k1w.jsonl
cat $OUT/r04_v14b_ab_k0_beside_k1w.jsonl
