import sys, os, time, numpy as np, torch, importlib
sys.path.insert(0, os.getcwd())
from _kmerdb_loader import import_kmerdb_amd
K = import_kmerdb_amd(); S = importlib.import_module('kmerdb_amd.synth')
import bench
dev = torch.device('cuda',0)
arr, names, counts, nk = bench.build_shard_db(K,S,1000,50,int(sys.argv[1]),18,20260929,dev,0,1)
n = arr['num_samples'].astype(np.int64); l = arr['num_local'].astype(np.int64)
upd = (n-l)*l + l*(l-1)//2
print('P',n.size,'mean n',n.mean(),'mean l',l.mean(),'updates',upd.sum(), 'frac nodes n>120', (n>120).mean(), 'upd frac n>120', upd[n>120].sum()/upd.sum(), 'n>64 upd frac', upd[n>64].sum()/upd.sum(), 'n>256', upd[n>256].sum()/upd.sum())
print('hist n', np.percentile(n,[50,90,99,99.9,100]), 'l', np.percentile(l,[50,90,99,99.9,100]))
d = bench.upload(K, arr, 1000, 18, 0)
M = torch.zeros(d.tri_size(), dtype=torch.int32, device=dev)
for name, fl in [('full',0),('skip scatter',2<<8),('skip flush',4<<8),('skip map+scatter',8<<8),('skip direct',16<<8),('global kernel',1)]:
    for _ in range(2):
        d.all2all_dense_device(M.data_ptr(), flags=fl)
    print(name, d.stats()['dominant_kernel_ms'], 'flushes', d.stats()['tile_flushes'])
