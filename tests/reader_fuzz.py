"""Corrupted .db images through the front-end reader (run as a process of its own by test_host_cpu.py: a reader that walks off its
mapping would take the test process down with it).  usage: reader_fuzz.py <good.db> <scratch.db> <seed> <count>"""
import os, sys, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
K = bench.import_kmerdb_amd()
src, out, seed, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
raw = open(src, "rb").read()
rng = np.random.default_rng(seed)
ok = bad = 0
for it in range(n):
    img = bytearray(raw)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        img = img[: int(rng.integers(0, len(img)))]
    elif kind == 1:
        for _ in range(int(rng.integers(1, 8))):
            img[int(rng.integers(0, len(img)))] = int(rng.integers(0, 256))
    elif kind == 2:                                  # a wild 32-bit or 64-bit field somewhere
        o = int(rng.integers(0, len(img) - 8))
        img[o:o + 8] = struct.pack("<Q", int(rng.integers(0, 1 << 62)))
    else:
        o = int(rng.integers(0, len(img) - 4))
        img[o:o + 4] = struct.pack("<I", int(rng.integers(0, 1 << 32)))
    open(out, "wb").write(bytes(img))
    os.environ["KMDB_LOAD_THREADS"] = str(int(rng.integers(1, 9)))
    for skip in (False, True):
        try:
            h = K.HostDB(out, skip_hashtables=skip)
            v = h.view_arrays()
            # whatever was accepted is internally consistent: every stream lies inside the data array
            words = (v["num_bits"].astype(np.uint64) + 127) // 128 * 2
            assert v["data_offset"].size == 0 or int((v["data_offset"] + words).max()) <= v["data"].size
            h.close()
            ok += 1
        except K.KmdbError:
            bad += 1
print("accepted", ok, "refused", bad)
