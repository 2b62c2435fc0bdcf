"""Randomised stress of the block-record pipeline against the HBM-atomics kernel (which the test-suite pins to the
oracle): random pattern forests, random block widths (even and odd), small and large sample counts (one block up to
hundreds of blocks: far more streams than a wave's 64 open chunks, so every reservation path runs: hit, crossing the end of a
chunk, eviction), short DFS slices (chain-table hand-over every 64 nodes) and pattern-stream slices.
usage: python profiles/r02_fuzz_stress.py [cases=60] [seed=1] [max_patterns=30000]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402
from test_gpu_parity import _random_forest    # noqa: E402

K = import_kmerdb_amd()
S = importlib.import_module("kmerdb_amd.synth")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAXP = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
rng_l2 = np.random.default_rng(7 + (int(sys.argv[2]) if len(sys.argv) > 2 else 1))     # (a generator of its own: the forests of a seed stay what they were)
bad = 0
joined = 0
for c in range(cases):
    N = int(rng.choice([2, 3, 31, 64, 65, 100, 257, 600, 1000, 1500, 2048, 3000, 7000, 12000]))
    P = int(rng.integers(5, MAXP))
    max_local = int(rng.choice([1, 2, 5, 40, 200, 900]))
    width = int(rng.choice([0, 0, 32, 33, 47, 50, 63, 64]))
    nseg = int(rng.choice([0, 64, 192]))
    chain = float(rng.choice([0.0, 0.0, 0.6, 0.9]))
    rowmode = int(rng.choice([0, 0, 1]))                   # 1: the many-streams path (per-block-row chunks + sort inside the rows) whatever the size
    wseg = int(rng.choice([0, 64, 128]))                   # wide nodes per run of the wide-node kernel (a chain seeded from the root path per run)
    k1w = int(rng.choice([0, 1, 3]))                       # waves of the wide-node kernel (few: every wave takes many runs)
    l2min = int(rng_l2.choice([11, 11, 24, 0]))            # second level above the block records (row mode, <= 256 blocks): threshold; 0 = off
    pat = _random_forest(rng, N, P, max_local, heavy_frac=float(rng.random()) * 0.5, zero_frac=float(rng.random()) * 0.5, chain_frac=chain)
    arr = S.to_view_arrays(pat)
    view = K.make_view(18, N, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"],
                       arr["last_sample_id"], arr["num_bits"], arr["data_offset"], arr["data"])
    for name, val in (("KMDB_BLOCK_WIDTH", width), ("KMDB_NSEG", nseg), ("KMDB_ROW_MODE", rowmode), ("KMDB_K1W_RUN", wseg), ("KMDB_K1W_WAVES", k1w)):
        if val:
            os.environ[name] = str(val)
        else:
            os.environ.pop(name, None)
    os.environ["KMDB_L2"] = "1" if l2min else "0"
    os.environ["KMDB_L2_MIN"] = str(l2min or 24)
    d = K.DeviceDB(view, device=0)
    ref = d.all2all_dense(flags=K.capi.FLAG_FORCE_GLOBAL_ATOMICS)
    got = d.all2all_dense(flags=K.capi.FLAG_NO_FALLBACK)
    st = d.stats()
    joined += st["n_joined"]
    ok = np.array_equal(got, ref) and st["path"] == K.capi.PATH_RECORDS
    got2 = d.all2all_dense(flags=K.capi.FLAG_NO_FALLBACK)          # cached grid sizes
    ok = ok and np.array_equal(got2, ref) and d.stats()["sized_call"] == 0
    acc = np.zeros_like(ref)
    for sh in range(3):
        acc += d.all2all_dense(shard=(sh, 3), flags=K.capi.FLAG_NO_FALLBACK)
    ok = ok and np.array_equal(acc, ref)
    d.close()
    if not ok:
        bad += 1
        print("MISMATCH case", c, "N", N, "P", P, "max_local", max_local, "width", width, "nseg", nseg, "chain", chain, "rowmode", rowmode, "wseg", wseg, "k1w", k1w, "l2min", l2min,
              "diff cells", int((got != ref).sum()), flush=True)
for name in ("KMDB_BLOCK_WIDTH", "KMDB_NSEG", "KMDB_ROW_MODE", "KMDB_K1W_RUN", "KMDB_K1W_WAVES", "KMDB_L2", "KMDB_L2_MIN"):
    os.environ.pop(name, None)
print("fuzz: %d cases, %d mismatches (%d nodes took the second level)" % (cases, bad, joined))
sys.exit(1 if bad else 0)
