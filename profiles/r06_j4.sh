#!/bin/bash
# Round 6, job 4: where the direct narrow kernel's extra 0.75 ms go (job 3: emit_narrow 1.29 -> 2.04 ms at C2), and how idle the machine is inside a call.
#   KMDB_K1N_DBG=1 no write-back of the tile, =2 no matrix-core step, =3 neither (timing only, results wrong); KMDB_NSEG: nodes per slice
#   (fewer write-backs); profiles/r06_overlap_probe.py: two calls side by side against two calls one after the other.
TAG=r06_j4
OUT=$PWD/gpurun_out
mkdir -p $OUT
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()}, d["roofline"].get("block_records_per_launch"))
except Exception as e: print("$name: no line", e)
PY
}
ab c2_direct "" KMDB_K1N_DIRECT=1
ab c2_noflush "" KMDB_K1N_DIRECT=1 KMDB_K1N_DBG=1
ab c2_nomfma "" KMDB_K1N_DIRECT=1 KMDB_K1N_DBG=2
ab c2_neither "" KMDB_K1N_DIRECT=1 KMDB_K1N_DBG=3
ab c2_nseg4096 "" KMDB_K1N_DIRECT=1 KMDB_NSEG=4096
ab c2_nseg8192 "" KMDB_K1N_DIRECT=1 KMDB_NSEG=8192
ab c2_chunks "" KMDB_K1N_DIRECT=0
KMDB_K1N_DIRECT=0 timeout 600 python profiles/r06_overlap_probe.py c2 10 2>/dev/null | tail -1 | tee $OUT/${TAG}_overlap_c2.txt
KMDB_K1N_DIRECT=0 timeout 600 python profiles/r06_overlap_probe.py c3part 10 2>/dev/null | tail -1 | tee $OUT/${TAG}_overlap_c3part.txt
