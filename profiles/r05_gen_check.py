# Round 5: the synthetic generator's short-genome forms (kmers_of over an (n, k) view, phase A's k-mer sets kept for phase B) on the GPU:
# equal to the loop forms and to the CPU's result, and what a sample costs now.   python profiles/r05_gen_check.py
import importlib
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402

import_kmerdb_amd()
S = importlib.import_module("kmerdb_amd.synth")
dev = torch.device("cuda", 0)
g = S.CladeGenomes(200, 50, 3000, seed=3, device=dev)
for k in (18, 25, 31):
    for f in (1.0, 0.1):
        for i in (0, 57, 199):
            c = g.sample(i)
            S._WINDOWS_AT_ONCE = 1 << 16
            a = S.kmers_of(c, k, f)
            S._WINDOWS_AT_ONCE = 0
            b = S.kmers_of(c, k, f)
            assert torch.equal(a, b), (k, f, i)
print("kmers_of: (n, k) form == loop form on the GPU")


def run(device, keep, at_once, n=2000):
    S._KEEP_KMERS, S._WINDOWS_AT_ONCE = keep, at_once
    if device.type == "cuda":
        torch.cuda.synchronize()
    t = time.time()
    _, pat = S.synth_database(n, 50, 700, k=25, fraction=0.1, seed=11, device=device)
    if device.type == "cuda":
        torch.cuda.synchronize()
    return pat, (time.time() - t) / n * 1e3


new, t_new = run(dev, 1 << 26, 1 << 16)
old, t_old = run(dev, 0, 0)
cpu, _ = run(torch.device("cpu"), 1 << 26, 1 << 16)
for key in new:
    for other in (old, cpu):
        a, b = new[key], other[key]
        if isinstance(a, torch.Tensor):
            assert torch.equal(a.cpu(), b.cpu()), key
        else:
            assert list(a) == list(b), key
print("synth_database (2000 samples x 700 bp, k=25 f=0.1): new forms == loop forms == CPU; %.2f ms per sample against %.2f" % (t_new, t_old))
