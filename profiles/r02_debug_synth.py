# debugging aid: one synthetic database (N samples, clade size, genome length) against the HBM-atomics kernel
import importlib, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["KMDB_VERBOSE"] = "1"
from _kmerdb_loader import import_kmerdb_amd
K = import_kmerdb_amd(); S = importlib.import_module('kmerdb_amd.synth')
N, cs, L = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 18
f = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
dev = torch.device('cuda', 0)
g, pat = S.synth_database(N, cs, L, k=k, fraction=f, seed=11, device=dev)
arr = S.to_view_arrays(pat)
view = K.make_view(k, N, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"], arr["last_sample_id"], arr["num_bits"], arr["data_offset"], arr["data"])
d = K.DeviceDB(view, device=0)
print("P", d.P, flush=True)
got = d.all2all_dense()
st = d.stats()
print({k_: st[k_] for k_ in ("path", "kernel_ms", "k0_ms", "k1n_ms", "k1g_ms", "k2_ms", "n_records", "n_wide", "n_chunks", "sum_pairs", "width")}, flush=True)
got2 = d.all2all_dense(); st = d.stats()
print("warm", {k_: st[k_] for k_ in ("kernel_ms", "k0_ms", "k1n_ms", "k1g_ms", "k2_ms")}, flush=True)
print("checksum ok:", int(got.astype(np.uint64).sum()) == st["sum_pairs"], flush=True)
if d.tri_size() < 3e8:
    ref = d.all2all_dense(flags=K.capi.FLAG_FORCE_GLOBAL_ATOMICS)
    print("== v1 global:", np.array_equal(got, ref), "diff", int((got != ref).sum()), flush=True)
