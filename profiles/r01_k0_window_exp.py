import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from _kmerdb_loader import import_kmerdb_amd
K = import_kmerdb_amd(); S = importlib.import_module('kmerdb_amd.synth')
import bench
dev = torch.device('cuda', 0)
arr, names, counts, nk = bench.build_shard_db(K, S, 1000, 50, int(sys.argv[1]), 18, 20260929, dev, 0, 1)
for win in sys.argv[2:]:
    os.environ['KMDB_K0_WINDOW'] = win
    d = bench.upload(K, arr, 1000, 18, 0)
    M = torch.zeros(d.tri_size(), dtype=torch.int32, device=dev)
    for _ in range(3): d.all2all_dense_device(M.data_ptr())
    d.all2all_dense_device(M.data_ptr(), flags=32 << 8)   # phase profile of the emit kernel (load, doubling, emit, chain)
    d.all2all_dense_device(M.data_ptr())
    for nm, fl in (('no record stores', 256 << 8), ('no emit section', 512 << 8), ('K2 no flush', 1024 << 8), ('K2 no row loop', 2048 << 8), ('K2 neither', 3072 << 8), ('K1 without wide-leaf split', 4096 << 8)):
        d.all2all_dense_device(M.data_ptr(), flags=fl); d.all2all_dense_device(M.data_ptr(), flags=fl)
        print('   ', nm, 'k1 %.3f k2 %.3f records %d' % (d.stats()['k1_ms'], d.stats()['k2_ms'], d.stats()['n_records']))
    for ev in ('KMDB_SKIP_K0A', 'KMDB_SKIP_K0B'):
        os.environ[ev] = '1'
        d.all2all_dense_device(M.data_ptr()); d.all2all_dense_device(M.data_ptr())
        print('   ', ev, 'k0 %.3f' % d.stats()['k0_ms'])
        del os.environ[ev]
    d.all2all_dense_device(M.data_ptr())
    st = d.stats()
    print('WIN', win, 'total %.3f k0 %.3f k1 %.3f k2 %.3f' % (st['kernel_ms'], st['k0_ms'], st['k1_ms'], st['k2_ms']), flush=True)
    d.close()
