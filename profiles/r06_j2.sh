#!/bin/bash
# Round 6, second job (prepared at the end of round 5, compiled for gfx950, never run): the short decode launch fetches what a node's
# decode needs — k0in word, stream offset, the first six units of its stream — in DFS order BEFORE the re-deal and hands it over through
# LDS (profiles/r05_k0_lds_prologue_prepared.diff; 30 VGPRs, 9984 B of LDS per workgroup, still 8 waves per SIMD).  Today the decoding
# thread fetches k0in[i], bitrel[i] and the units in permuted order: three dependent gathers (k0<short>: 67 % of its wave time waiting,
# profiles/r05_close_c3part_sq_counters.md).   git apply profiles/r05_k0_lds_prologue_prepared.diff && python -c "import __graft_entry__ as g; g.build()"   first.
TAG=r06_j2
OUT=$PWD/gpurun_out
mkdir -p $OUT
KMDB_K0_PRO=1 timeout 900 python -m pytest tests -m gpu -q -x --durations=5 -k "all2all_dense_bit_exact or synthetic_databases or random_forests or randomised_stress or pools_too_small or degenerate or second_level or many_samples or touch_every_block or 10000-50-400" > $OUT/${TAG}_tests_sel.log 2>&1; tail -9 $OUT/${TAG}_tests_sel.log | cut -c1-200
ab() {  # name, workload args, env...
  name=$1; shift; wl=$1; shift
  env "$@" timeout 400 python bench.py $wl --no-cpu-baseline --no-extra --steps 20 --warmup 3 > $OUT/${TAG}_ab_$name.json 2> $OUT/${TAG}_ab_$name.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_ab_$name.json")); print("$name", round(d["ms_per_step"],3), {k:round(v,3) for k,v in d["roofline"]["per_kernel_ms"].items()})
except Exception as e: print("$name: no line", e)
PY
}
ab c2_gather_a "" KMDB_K0_PRO=0
ab c2_lds_a "" KMDB_K0_PRO=1
ab c2_gather_b "" KMDB_K0_PRO=0
ab c2_lds_b "" KMDB_K0_PRO=1
ab c3_gather "--workload c3part" KMDB_K0_PRO=0
ab c3_lds "--workload c3part" KMDB_K0_PRO=1
