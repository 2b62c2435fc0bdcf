"""Ablation driver (not part of the product): times the kernel variants on one synthetic DB.
usage: python profiles/r01_ablation_exp.py <genome_length>"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402
K = import_kmerdb_amd()
S = importlib.import_module('kmerdb_amd.synth')
import bench  # noqa: E402

dev = torch.device('cuda', 0)
arr, names, counts, nk = bench.build_shard_db(K, S, 1000, 50, int(sys.argv[1]), 18, 20260929, dev, 0, 1)
n = arr['num_samples'].astype(np.int64)
l = arr['num_local'].astype(np.int64)
upd = (n - l) * l + l * (l - 1) // 2
print('P', n.size, 'mean n', n.mean(), 'mean l', l.mean(), 'updates', upd.sum(), 'upd frac n>64', upd[n > 64].sum() / upd.sum(),
      'n>120', upd[n > 120].sum() / upd.sum(), 'n>256', upd[n > 256].sum() / upd.sum(), 'max n', n.max())
d = bench.upload(K, arr, 1000, 18, 0)
M = torch.zeros(d.tri_size(), dtype=torch.int32, device=dev)
for name, fl in [('v3 K0+K1par+K2', 0), ('v2 seq emit', 8), ('v2 profiled', 8 | (32 << 8)), ('v2 K2 without popcount items', 64 << 8), ('v2 K2 without scatter items', 128 << 8), ('v2 K1 only (no apply)', 2 << 8), ('v1 tile', 4), ('v1 tile, no scatter', 4 | (2 << 8)),
                 ('v1 tile, decode+stack only', 4 | (8 << 8)), ('generic HBM atomics', 1), ('LDS stack + HBM atomics', 2)]:
    for _ in range(2):
        d.all2all_dense_device(M.data_ptr(), flags=fl)
    st = d.stats()
    print('%-28s total %.3f ms  dominant %.3f  k0 %.3f k1 %.3f k2 %.3f  records %d flushes %d' % (
        name, st['kernel_ms'], st['dominant_kernel_ms'], st['k0_ms'], st['k1_ms'], st['k2_ms'], st['n_records'], st['tile_flushes']))
