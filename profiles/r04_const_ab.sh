#!/bin/bash
# Same-box A/B of compile-time constants of a2a_blocks.hip: every argument is "NAME=value[,NAME=value...][@ENV=val ENV2=val]"; the library
# is rebuilt on the box per variant and profiles/r04_ab.py times the warm calls on databases generated once (kept in shared memory).
#   WL="c3part c2" bash profiles/r04_const_ab.sh "K2_PF=1" "K2_PF=3,K2S_MIN_WAVES=2,K2A_MIN_WAVES=2" "@KMDB_K1N_FUSED=0"
cd "$(dirname "$0")/.."
cp kmer-db_amd/csrc/a2a_blocks.hip /tmp/a2a_blocks.orig
for v in "" "$@"; do
  cp /tmp/a2a_blocks.orig kmer-db_amd/csrc/a2a_blocks.hip
  consts=${v%%@*}; envs=""; [[ "$v" == *@* ]] && envs=${v#*@}
  for kv in ${consts//,/ }; do
    name=${kv%%=*}; val=${kv#*=}
    sed -i -E "s/(constexpr (uint32_t|int) ([A-Z0-9_]+ = [0-9]+, )*)$name = [0-9]+/\1$name = $val/" kmer-db_amd/csrc/a2a_blocks.hip
  done
  make -C kmer-db_amd -j8 > /dev/null 2>&1 || echo "BUILD FAILED for $v"
  for w in ${WL:-c3part c2}; do
    echo -n "[${v:-as committed}] $w: "
    python profiles/r04_ab.py $w "$envs" 2>/dev/null | tail -1
  done
done
cp /tmp/a2a_blocks.orig kmer-db_amd/csrc/a2a_blocks.hip
make -C kmer-db_amd -j8 > /dev/null 2>&1
