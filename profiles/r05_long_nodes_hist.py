"""Round 5: which nodes does the long decode launch (k0_decode_kernel<true>) spend its time on?  Histogram of the local lists the
upload sends there (more than 48 ids or more than 128 stream bits), by stream length: nodes, ids, stream bits per class.
Run from the repo root on the GPU box:  python profiles/r05_long_nodes_hist.py c2|c3part"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import bench

name = sys.argv[1]
wl = bench.WORKLOADS[name]
K = bench.import_kmerdb_amd()
import importlib
S = importlib.import_module("kmerdb_amd.synth")
dev = torch.device("cuda", 0)
arr, names, counts, nk, items = bench.build_db(K, S, wl["samples"], wl["clade_size"], wl["length"], 18, 1, dev, 0, 1)
l = arr["num_local"].astype(np.int64)
b = arr["num_bits"].astype(np.int64)
P = l.size
long_ = (l > 48) | (b > 128)
print("%s: %d nodes, l == 0: %.1f %%, l == 1: %.1f %%, long: %d (%.2f %%), their ids %.1f M of %.1f M, their stream bits %.1f M of %.1f M" % (
    name, P, 100.0 * (l == 0).mean(), 100.0 * (l == 1).mean(), long_.sum(), 100.0 * long_.mean(), l[long_].sum() / 1e6, l.sum() / 1e6, b[long_].sum() / 1e6, b.sum() / 1e6))
edges = [0, 129, 257, 513, 1025, 2049, 4097, 8193, 16385, 1 << 40]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = long_ & (b >= lo) & (b < hi)
    if m.any():
        # decode steps ~ codes that are not "0" + runs: stream bits beyond one per id is the launch's own estimate of work
        print("  bits [%6d, %6d): %9d nodes  ids %8.2f M  bits %8.2f M  bits beyond one per id %8.2f M   mean l %.0f" % (
            lo, min(hi, 1 << 20), m.sum(), l[m].sum() / 1e6, b[m].sum() / 1e6, (b[m] - (l[m] - 1)).clip(0).sum() / 1e6, l[m].mean()))
sh = ~long_ & (l > 1)
print("  short launch: %d nodes with l > 1, ids %.1f M, bits %.1f M, mean l %.1f, mean bits %.1f" % (sh.sum(), l[sh].sum() / 1e6, b[sh].sum() / 1e6, l[sh].mean(), b[sh].mean()))
for lo, hi in ((2, 3), (3, 5), (5, 9), (9, 17), (17, 33), (33, 49)):
    m = sh & (l >= lo) & (l < hi)
    print("    l in [%2d, %2d): %9d nodes (%.1f %% of all)  mean bits %.1f" % (lo, hi, m.sum(), 100.0 * m.sum() / P, b[m].mean() if m.any() else 0))
