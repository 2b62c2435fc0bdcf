#!/bin/bash
# same-box A/B of the wide kernel's row length (LDS per wave) and wave count: rebuilds the library on the box per variant
#   bash profiles/r02_k1g_ab.sh
cd "$(dirname "$0")/.."
for v in "8 8192" "12 8192" "12 4096" "8 4096" "10 8192"; do
  set -- $v
  sed -i "s/^constexpr uint32_t K1G_ROW = [0-9]*; */constexpr uint32_t K1G_ROW = $1;  /; s/^constexpr uint32_t K1G_MAX_WAVES = [0-9]*;/constexpr uint32_t K1G_MAX_WAVES = $2;/" kmer-db_amd/csrc/a2a_blocks.hip
  make -C kmer-db_amd -j8 > /dev/null 2>&1
  for w in c2 c3part; do
    python bench.py --workload $w --no-cpu-baseline --steps 5 > /tmp/ab.json 2> /dev/null
    python3 -c "
import json
c=json.load(open('/tmp/ab.json')); print('row $1 waves $2 $w', round(c['ms_per_step'],3), {k:round(v,2) for k,v in c['roofline']['per_kernel_ms'].items()}, c['roofline']['wide_nodes_climbing'])"
  done
done
