#!/usr/bin/env python3
"""How much of the call is idle machine?  Two handles of the SAME database (own streams, own pools), timed one after the other and side by
side from two host threads.  If two concurrent calls take much less than twice one call, the kernels of a call leave the machine
under-used and a call pipelined over halves of its pattern stream (decode of one half beside the sort of the other ...) would gain about that.

    python profiles/r06_overlap_probe.py [workload] [calls]
"""
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    w = bench.WORKLOADS[wl]
    K = import_kmerdb_amd()
    torch.cuda.set_device(0)
    arr, names, counts, nk, items = bench.generate_in_child(0, n_samples=w["samples"], clade_size=w["clade_size"], length=w["length"], k=18, seed=20260929,
                                                           rank=0, world=1, progress=None)
    dbs = [bench.upload(K, arr, w["samples"], 18, 0)[0] for _ in range(2)]
    cells = dbs[0].tri_size()
    Ms = [torch.zeros(cells, dtype=torch.int32, device="cuda") for _ in range(2)]
    for d, M in zip(dbs, Ms):
        for _ in range(3):
            d.all2all_dense_device(M.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(Ms[0], Ms[1])

    def run(i, n):
        for _ in range(n):
            dbs[i].all2all_dense_device(Ms[i].data_ptr())

    t0 = time.perf_counter()
    run(0, calls)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / calls * 1e3
    t0 = time.perf_counter()
    for _ in range(calls):
        run(0, 1)
        run(1, 1)
    torch.cuda.synchronize()
    seq = (time.perf_counter() - t0) / calls * 1e3
    th = [threading.Thread(target=run, args=(i, calls)) for i in range(2)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    con = (time.perf_counter() - t0) / calls * 1e3
    assert torch.equal(Ms[0], Ms[1])
    # the same with the second chain half a call behind the first: decode / narrow emit of one beside wide emit / sort / apply of the other
    stag = []
    for frac in (0.35, 0.5, 0.65):
        def run_late(i, n, delay):
            if delay:
                time.sleep(delay)
            run(i, n)
        th = [threading.Thread(target=run_late, args=(i, calls, one * 1e-3 * frac * i)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        stag.append((frac, (time.perf_counter() - t0 - one * 1e-3 * frac) / calls * 1e3))
    assert torch.equal(Ms[0], Ms[1])
    print("%s: staggered starts (fraction of a call, ms per pair of calls): %s" % (wl, ", ".join("%.2f: %.3f" % x for x in stag)))
    print("%s: one call %.3f ms; two calls one after the other %.3f ms; two calls side by side (two host threads, own streams and pools) %.3f ms = %.2f x one call" %
          (wl, one, seq, con, con / one))


if __name__ == "__main__":
    main()
