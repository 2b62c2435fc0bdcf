#!/bin/bash
# same-box A/B of compile-time constants of a2a_blocks.hip: every argument is "NAME=value[,NAME=value...]"; the library is rebuilt
# on the box per variant, c2 (and with C3=1 also c3part) timed.   bash profiles/r02_const_ab.sh "CS_BLOCKS=1536" "CS_BLOCKS=3072"
cd "$(dirname "$0")/.."
cp kmer-db_amd/csrc/a2a_blocks.hip /tmp/a2a_blocks.orig
for v in "" "$@"; do
  cp /tmp/a2a_blocks.orig kmer-db_amd/csrc/a2a_blocks.hip
  for kv in ${v//,/ }; do
    name=${kv%%=*}; val=${kv#*=}
    sed -i -E "s/(constexpr (uint32_t|int) ([A-Z0-9_]+ = [0-9]+, )*)$name = [0-9]+/\1$name = $val/" kmer-db_amd/csrc/a2a_blocks.hip
  done
  make -C kmer-db_amd -j8 > /dev/null 2>&1
  for w in c2 ${C3:+c3part}; do
    python bench.py --workload $w --no-cpu-baseline --no-extra --steps 5 > /tmp/ab.json 2> /dev/null
    python3 -c "
import json
c=json.load(open('/tmp/ab.json')); print('%-28s %-6s' % ('${v:-(as committed)}', '$w'), round(c['ms_per_step'],3), {k:round(v,2) for k,v in c['roofline']['per_kernel_ms'].items()})"
  done
done
cp /tmp/a2a_blocks.orig kmer-db_amd/csrc/a2a_blocks.hip
