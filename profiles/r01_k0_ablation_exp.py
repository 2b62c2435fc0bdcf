import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from _kmerdb_loader import import_kmerdb_amd
K = import_kmerdb_amd(); S = importlib.import_module('kmerdb_amd.synth')
import bench
dev = torch.device('cuda', 0)
arr, names, counts, nk = bench.build_shard_db(K, S, 1000, 50, int(sys.argv[1]), 18, 20260929, dev, 0, 1)
d = bench.upload(K, arr, 1000, 18, 0)
M = torch.zeros(d.tri_size(), dtype=torch.int32, device=dev)
for _ in range(3): d.all2all_dense_device(M.data_ptr())
for dbg in sys.argv[2:]:
    os.environ['KMDB_K0_DBG'] = dbg
    for ev in ('', 'KMDB_SKIP_K0A', 'KMDB_SKIP_K0B'):
        if ev: os.environ[ev] = '1'
        d.all2all_dense_device(M.data_ptr()); d.all2all_dense_device(M.data_ptr())
        print('dbg', dbg, ev, 'k0 %.3f' % d.stats()['k0_ms'], flush=True)
        if ev: del os.environ[ev]
