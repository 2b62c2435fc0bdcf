#!/bin/bash
# Why did 16 instead of 14 waves per CU in the wide kernel cost the row sort + apply 1 ms (r04_v11)?  Every wave leaves one partly filled chunk per
# block row behind.  (1) waves of the wide kernel 1024 .. 4096, same build; (2) 128-record chunks, 32 / 128 chunks per sort job
OUT=$PWD/gpurun_out; mkdir -p $OUT
python profiles/r04_ab.py c3part "" "KMDB_K1W_WAVES=4096" "KMDB_K1W_WAVES=3072" "KMDB_K1W_WAVES=2560" "KMDB_K1W_WAVES=2048" "KMDB_K1W_WAVES=1536" "KMDB_K1W_WAVES=1024" 2>/dev/null > $OUT/r04_v12_ab_k1w_waves.jsonl
cat $OUT/r04_v12_ab_k1w_waves.jsonl
WL="c3part c2" bash profiles/r04_const_ab.sh "CH_SHIFT=7" "RS_JOB_CHUNKS=32" "RS_JOB_CHUNKS=128" "CH_SHIFT=7,RS_JOB_CHUNKS=128" > $OUT/r04_v12_const_ab_chunks.txt 2>&1
cat $OUT/r04_v12_const_ab_chunks.txt
