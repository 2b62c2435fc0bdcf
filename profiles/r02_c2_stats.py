# Node statistics of the benchmark database + upload phase timings (KMDB_VERBOSE):
#   python profiles/r02_c2_stats.py <genome_len> [<samples> [<clade>]]
import importlib, os, sys, time, json, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["KMDB_VERBOSE"] = "1"
from _kmerdb_loader import import_kmerdb_amd
K = import_kmerdb_amd(); S = importlib.import_module('kmerdb_amd.synth')
import bench
L = int(sys.argv[1]); NS = int(sys.argv[2]) if len(sys.argv) > 2 else 1000; CL = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device('cuda', 0)
arr, names, counts, nk = bench.build_shard_db(K, S, NS, CL, L, 18, 20260929, dev, 0, 1)
P = arr["num_kmers"].size
par = torch.from_numpy(arr["parent_id"]).to(dev)
n = torch.from_numpy(arr["num_samples"].astype(np.int64)).to(dev)
l = torch.from_numpy(arr["num_local"].astype(np.int64)).to(dev)
nb = torch.from_numpy(arr["num_bits"].astype(np.int64)).to(dev)
w = torch.from_numpy(arr["num_kmers"]).to(dev)
out = {"P": P, "N": NS, "L": L}
# depth by pointer jumping
depth = torch.ones(P, dtype=torch.int32, device=dev); anc = par.clone()
rounds = 0
while True:
    act = anc >= 0
    if not bool(act.any()): break
    idx = anc[act]
    depth[act] += depth[idx]
    anc[act] = anc[idx]
    rounds += 1
out["max_depth"] = int(depth.max()); out["mean_depth"] = float(depth.float().mean()); out["jump_rounds"] = rounds
haschild = torch.zeros(P, dtype=torch.bool, device=dev); haschild[par[par >= 0]] = True
out["leaf_frac"] = float((~haschild).float().mean())
out["w0_frac"] = float((w == 0).float().mean()); out["w1_frac"] = float((w == 1).float().mean())
out["w_ge128_frac"] = float((w >= 128).float().mean())
out["roots"] = int((par < 0).sum())
for name, t in (("n", n), ("l", l), ("nbits", nb), ("depth", depth.long())):
    q = torch.tensor([0.5, 0.9, 0.99, 0.999], device=dev)
    s = t.float()
    out[name] = {"mean": float(s.mean()), "max": int(t.max()), "q50_90_99_999": [float(x) for x in torch.quantile(s[torch.randint(0, P, (4_000_000,), device=dev)], q)]}
out["l_le1_frac"] = float((l <= 1).float().mean()); out["l_gt32_frac"] = float((l > 32).float().mean())
out["parent_gap_mean"] = float((torch.arange(P, device=dev) - par)[par >= 0].float().mean())
del par, n, l, nb, w, depth, anc, haschild
torch.cuda.empty_cache()
t0 = time.time()
d = bench.upload(K, arr, NS, 18, 0)
out["upload_s"] = time.time() - t0
M = torch.zeros(d.tri_size(), dtype=torch.int32, device=dev)
for _ in range(3): d.all2all_dense_device(M.data_ptr())
st = d.stats(); out["stats"] = st
print(json.dumps(out), flush=True)
