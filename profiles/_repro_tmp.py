import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _kmerdb_loader import import_kmerdb_amd
from test_gpu_parity import _random_forest
K = import_kmerdb_amd(); S = importlib.import_module("kmerdb_amd.synth")
rng = np.random.default_rng(7)
for c in range(9):
    N = int(rng.choice([2, 3, 31, 64, 65, 100, 257, 600, 1000, 1500, 2048])); P = int(rng.integers(5, 30000))
    max_local = int(rng.choice([1, 2, 5, 40, 200])); width = int(rng.choice([0, 0, 32, 33, 47, 50, 63, 64]))
    if width and (N + width - 1) // width > 32: width = 0
    pat = _random_forest(rng, N, P, max_local, heavy_frac=float(rng.random()) * 0.5, zero_frac=float(rng.random()) * 0.5)
arr = S.to_view_arrays(pat)
view = K.make_view(18, N, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"], arr["last_sample_id"], arr["num_bits"], arr["data_offset"], arr["data"])
os.environ["KMDB_VERBOSE"] = "1"
d = K.DeviceDB(view, device=0)
ref = d.all2all_dense(flags=K.capi.FLAG_FORCE_GLOBAL_ATOMICS)
got = d.all2all_dense()
diff = np.nonzero(got != ref)[0]
print("N", N, "P", P, "ndiff", diff.size, "of", ref.size)
w = arr["num_kmers"]; print("weights: ==1", (w==1).sum(), "2..127", ((w>1)&(w<128)).sum(), ">=128", (w>=128).sum(), "max", w.max())
for i in diff[:12]:
    r = int((1 + np.sqrt(1 + 8 * i)) // 2); cc = int(i - r * (r - 1) // 2)
    print("cell", r, cc, "got", int(got[i]), "ref", int(ref[i]), "delta", int(got[i]) - int(ref[i]))
