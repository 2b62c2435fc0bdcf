# per-kernel times of the block-record pipeline on the benchmark model for a list of environment settings:
#   python profiles/r02_knob_exp.py <genome_len> "KMDB_NSEG=4096" "KMDB_NSEG=8192 KMDB_X=1" ...
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from _kmerdb_loader import import_kmerdb_amd
K = import_kmerdb_amd(); S = importlib.import_module('kmerdb_amd.synth')
import bench
dev = torch.device('cuda', 0)
arr, names, counts, nk, _ = bench.build_db(K, S, 1000, 50, int(sys.argv[1]), 18, 20260929, dev, 0, 1)
for setting in [""] + sys.argv[2:]:
    for kv in setting.split():
        k, v = kv.split("="); os.environ[k] = v
    d, up = bench.upload(K, arr, 1000, 18, 0)
    M = torch.zeros(d.tri_size(), dtype=torch.int32, device=dev)
    for _ in range(4): d.all2all_dense_device(M.data_ptr())
    st = d.stats()
    ok = int(M.to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) == st["sum_pairs"]
    print('%-40s upload %.2f  call %.3f = k0 %.3f k1n %.3f k1g %.3f k2 %.3f  chunks %d  checksum %s' % (
        setting or "(default)", up, st['kernel_ms'], st['k0_ms'], st['k1n_ms'], st['k1g_ms'], st['k2_ms'], st['n_chunks'], ok), flush=True)
    d.close()
    for kv in setting.split():
        os.environ.pop(kv.split("=")[0], None)
