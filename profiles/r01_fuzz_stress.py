"""Randomised stress of the block-record pipeline against the v1 kernels (which the test-suite pins to the oracle):
random pattern forests, random block widths (even and odd), both front halves.
usage: python profiles/r01_fuzz_stress.py [cases=60] [seed=1] [max_patterns=30000]"""
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
bad = 0
for c in range(cases):
    N = int(rng.choice([2, 3, 31, 64, 65, 100, 257, 600, 1000, 1500, 2048]))
    P = int(rng.integers(5, MAXP))
    max_local = int(rng.choice([1, 2, 5, 40, 200]))
    width = int(rng.choice([0, 0, 32, 33, 47, 50, 63, 64]))
    if width and (N + width - 1) // width > 32:
        width = 0
    pat = _random_forest(rng, N, P, max_local, heavy_frac=float(rng.random()) * 0.5, zero_frac=float(rng.random()) * 0.5)
    arr = S.to_view_arrays(pat)
    view = K.make_view(18, N, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"],
                       arr["last_sample_id"], arr["num_bits"], arr["data_offset"], arr["data"])
    if width:
        os.environ["KMDB_BLOCK_WIDTH"] = str(width)
    else:
        os.environ.pop("KMDB_BLOCK_WIDTH", None)
    d = K.DeviceDB(view, device=0)
    ref = d.all2all_dense(flags=K.capi.FLAG_FORCE_GLOBAL_ATOMICS)
    got = d.all2all_dense(flags=int(os.environ.get('FUZZ_FLAGS', '0')))
    ok = np.array_equal(got, ref) and (d.stats()["n_records"] > 0 or arr["num_samples"].max() > 1024 or P < 3)
    dseq = K.DeviceDB(view, device=0, flags=K.capi.FLAG_FORCE_SEQ_EMIT)
    ok2 = np.array_equal(dseq.all2all_dense(), ref)
    dseq.close(); d.close()
    if not (np.array_equal(got, ref) and ok2):
        bad += 1
        print("MISMATCH case", c, "N", N, "P", P, "max_local", max_local, "width", width, flush=True)
print("fuzz: %d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
