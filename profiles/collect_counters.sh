#!/bin/bash
# Per-kernel hardware counters of one bench.py command on the GPU box (run from the repo root):
#   BENCH_ARGS="--workload c3part" bash profiles/collect_counters.sh <tag> [groups...]
# Every group is one rocprofv3 --pmc pass (8 SQ slots / 4 TCC slots per pass; never combined with another trace domain than
# --kernel-trace).  The per-dispatch CSVs are reduced to one table per kernel by profiles/summarize_counters.py:
#   gpurun_out/<tag>_counters.json  and  gpurun_out/<tag>_counters.md
set -u
TAG=${1:-rXX}
shift
GROUPS_WANTED=${*:-"sq1 sq2 sq3 fetch write"}
OUT=$PWD/gpurun_out
R=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
BA=${BENCH_ARGS:-}
# counters only for the engine's kernels (the synthetic generator's thousands of torch kernels would be serialised too)
KRE=${KERNEL_REGEX:-"k0_|k1n_|k1w_|k2_|k2d_|k2j_|l2_|cs_|rs_|rg_|ct_|wide_|wrun_|n2a_|d2_|row_nnz|row_compact|lay_|width_estimate|pair_estimate|iota_u32"}
declare -A G
G[sq1]="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS"
G[sq2]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM"
G[sq3]="SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT"
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
G[l2]="TCC_HIT_sum TCC_MISS_sum"
FILES=""
for g in $GROUPS_WANTED; do
  C=${G[$g]}
  rm -rf /tmp/pmc_$g
  ( cd /tmp && rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$KRE" --output-format csv -d /tmp/pmc_$g -- python $R/bench.py $BA --no-cpu-baseline --steps 2 --warmup 1 > /tmp/pmc_$g.json 2> /tmp/pmc_$g.err )
  f=""                                                                 # the bench process' file, not its generator child's: the one with the engine's kernels
  for c in $(find /tmp/pmc_$g -name '*counter_collection.csv'); do
    if grep -q -E "k0_decode_kernel|n2a_|d2_|row_nnz_kernel" $c; then f=$c; fi
  done
  if [ -n "$f" ]; then cp $f /tmp/pmc_$g.csv; FILES="$FILES /tmp/pmc_$g.csv"; else echo "group $g: no counter file"; tail -5 /tmp/pmc_$g.err; fi
done
python profiles/summarize_counters.py $TAG $OUT $FILES
