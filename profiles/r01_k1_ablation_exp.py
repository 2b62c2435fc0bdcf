# K1 (emit kernel) timing experiments: python profiles/r01_k1_ablation_exp.py <genome_len> <flag> [<flag> ...]
# flags are the debug bits of kmdb_opts.flags >> 8 (512 = no emit section, 8192 = ignore extra pairs, ...)
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
for fl in sys.argv[2:]:
    f = int(fl) << 8
    for ev in ('', 'KMDB_SKIP_K1N', 'KMDB_SKIP_K1W'):
        if ev: os.environ[ev] = '1'
        d.all2all_dense_device(M.data_ptr(), flags=f); d.all2all_dense_device(M.data_ptr(), flags=f)
        st = d.stats()
        print('dbg', fl, ev, 'k0 %.3f k1 %.3f k2 %.3f records %d' % (st['k0_ms'], st['k1_ms'], st['k2_ms'], st['n_records']), flush=True)
        if ev: del os.environ[ev]
