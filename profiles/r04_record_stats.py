#!/usr/bin/env python3
"""Where the block records come from: emitting nodes (w > 0, n >= 2) of a synthetic clade database by the number of blocks of
their full list, and the records (c (c + 1) / 2 per node) every class contributes.  CPU or GPU torch; the shape of the
distribution does not depend on the genome length.

    python profiles/r04_record_stats.py [samples] [length] [width] [device]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    dev = sys.argv[4] if len(sys.argv) > 4 else "cpu"
    import_kmerdb_amd()
    import importlib
    S = importlib.import_module("kmerdb_amd.synth")
    g = S.CladeGenomes(n, 50, L, seed=20260929, device=dev)
    pat = S.build_patterns(lambda i: S.kmers_of(g.sample(i), 18), n, dev)
    P = pat["num_kmers"].numel()
    par, lp, ids, l, w, ns = pat["parent"], pat["local_ptr"], pat["local_ids"], pat["num_local"], pat["num_kmers"], pat["num_samples"]
    blk = ids // width
    # local list: first block, last block, distinct blocks
    owner = torch.repeat_interleave(torch.arange(P, device=ids.device), l)
    newb = torch.ones_like(blk)
    if blk.numel() > 1:
        same_owner = owner[1:] == owner[:-1]
        newb[1:] = torch.where(same_owner & (blk[1:] == blk[:-1]), 0, 1)
    cl = torch.zeros(P, dtype=torch.int64, device=ids.device).index_add_(0, owner, newb)
    has = l > 0
    fb = torch.full((P,), -1, dtype=torch.int64, device=ids.device)
    lb = torch.full((P,), -2, dtype=torch.int64, device=ids.device)
    fb[has] = blk[lp[:-1][has]]
    lb[has] = blk[lp[1:][has] - 1]
    # blocks of the full list, parents before children (parent < child in creation order): level by level
    depth = torch.zeros(P, dtype=torch.int64, device=ids.device)
    c = cl.clone()
    lastb = lb.clone()
    order = torch.arange(P, device=ids.device)
    # creation order guarantees parent < child: a sequential sweep in chunks by depth
    d = torch.zeros(P, dtype=torch.int64, device=ids.device)
    hp = par >= 0
    cur = hp.clone()
    it = 0
    # depth by repeated relaxation
    dd = torch.zeros(P, dtype=torch.int64, device=ids.device)
    while True:
        nd = torch.where(hp, dd[par.clamp(min=0)] + 1, torch.zeros_like(dd))
        if torch.equal(nd, dd):
            break
        dd = nd
        it += 1
    maxd = int(dd.max())
    for lev in range(1, maxd + 1):
        sel = (dd == lev).nonzero().squeeze(1)
        p = par[sel]
        seam = (fb[sel] == lastb[p]).to(torch.int64)
        c[sel] = c[p] + cl[sel] - seam
        lastb[sel] = torch.where(has[sel], lb[sel], lastb[p])
    emit = (w > 0) & (ns >= 2)
    ce = c[emit]
    recs = ce * (ce + 1) // 2
    tot = int(recs.sum())
    print("samples %d length %d width %d: %d patterns, %d emitting, %d records (%.2f per pattern), max depth %d" % (n, L, width, P, int(emit.sum()), tot, tot / P, maxd))
    edges = [1, 2, 3, 4, 6, 8, 11, 16, 24, 32, 48, 64, 96, 128, 1 << 30]
    print("%12s %12s %8s %14s %8s" % ("blocks", "nodes", "%nodes", "records", "%records"))
    for a, b in zip(edges[:-1], edges[1:]):
        m = (ce >= a) & (ce < b)
        print("%5d..%-5s %12d %8.2f %14d %8.2f" % (a, "" if b > 1 << 20 else b - 1, int(m.sum()), 100.0 * int(m.sum()) / max(1, ce.numel()), int(recs[m].sum()), 100.0 * int(recs[m].sum()) / max(1, tot)))
    # K0's long launch: nodes with more than 32 local ids or more than 128 stream bits; the ones whose local list spans 64 ids or more are
    # decoded twice
    delta = torch.zeros_like(ids)
    if ids.numel() > 1:
        delta[1:] = ids[1:] - ids[:-1]
    first = torch.ones_like(ids, dtype=torch.bool)
    if ids.numel() > 1:
        first[1:] = owner[1:] != owner[:-1]
    bl = torch.frexp(delta.clamp(min=1).to(torch.float64))[1].to(torch.int64)
    clen = torch.where(first, torch.zeros_like(bl), 2 * bl - 1)
    nbits = torch.zeros(P, dtype=torch.int64, device=ids.device).index_add_(0, owner, clen)
    span = torch.zeros(P, dtype=torch.int64, device=ids.device)
    span[has] = ids[lp[1:][has] - 1] - ids[lp[:-1][has]]
    long_ = (l > 32) | (nbits > 128)
    print("long nodes (l > 32 or bits > 128): %d = %.1f %% of the nodes; of them span >= 64 (two passes): %.1f %%; mean l of the long nodes %.1f, mean stream bits %.1f; "
          "short nodes with span >= 64: %.2f %% of all nodes" % (int(long_.sum()), 100.0 * float(long_.double().mean()), 100.0 * float((span[long_] >= 64).double().mean()),
                                                               float(l[long_].double().mean()), float(nbits[long_].double().mean()), 100.0 * float(((~long_) & (span >= 64)).double().mean())))
    nonone = torch.zeros(P, dtype=torch.int64, device=ids.device).index_add_(0, owner, ((delta > 1) & ~first).to(torch.int64))
    print("codes other than '0' per long node: mean %.1f; per short node with l > 1: mean %.2f" % (float(nonone[long_].double().mean()), float(nonone[(~long_) & (l > 1)].double().mean())))
    # weights of the emitting nodes
    we = w[emit]
    print("weights: mean %.2f, >=128: %d of %d; w==1: %.1f %%" % (float(we.double().mean()), int((we >= 128).sum()), we.numel(), 100.0 * float((we == 1).double().mean())))
    # distinct (blocks-set) reuse: how many wide nodes share the parent's block count (children adding no block)
    wide = emit & (c > 2)
    print("wide emitting nodes: %d; their parents wide too: %.1f %%" % (int(wide.sum()), 100.0 * float((c[par.clamp(min=0)][wide] > 2).double().mean())))


if __name__ == "__main__":
    main()
