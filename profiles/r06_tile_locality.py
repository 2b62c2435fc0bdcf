#!/usr/bin/env python3
"""VERDICT round 5, "next round" 1: measure FIRST how clustered the block records are in the DFS stream.

For a synthetic clade database (same generator and seed as bench.py's workloads, shorter genomes: the tree keeps its shape, every
region of it just holds fewer nodes — so what is counted here is a LOWER bound of the locality at full length) the script lays the
pattern tree out in the engine's DFS order and counts, on the CPU, no engine involved:

  narrow stream   per slice of 2048 nodes (one wave of the narrow kernel): distinct first blocks of the emitting nodes with at most two
                  blocks, and the share of their first-block records (X, X) that belongs to the slice's most frequent block
  wide stream     the emitting nodes with 3 .. L2_MIN-1 blocks in DFS order, in batches of 64 (one wave step of the wide kernel): every
                  node contributes one record to each tile (a, b), a >= b, of its block list; per batch the tiles are ranked by the
                  number of contributing nodes.  Reported: share of the records that lie in tiles with >= T contributors in their batch
                  (an in-wave MFMA step over those tiles needs no record in HBM), and the share covered by the K most frequent tiles of a
                  run of 4 batches (a small set of tiles resident in LDS)

    python profiles/r06_tile_locality.py [samples] [length] [width] [batches sampled]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402


def dfs_order(par):
    """pre-order index of every node; children of a parent in pid order; roots in pid order (the engine's layout.hip order)"""
    P = par.size
    depth = np.zeros(P, np.int32)
    # parent < child in creation order: depth by one forward sweep, vectorised level by level
    order = np.arange(P)
    hp = par >= 0
    d = np.zeros(P, np.int32)
    while True:
        nd = np.where(hp, d[np.maximum(par, 0)] + 1, 0).astype(np.int32)
        if np.array_equal(nd, d):
            break
        d = nd
    depth = d
    maxd = int(depth.max())
    size = np.ones(P, np.int64)
    for lev in range(maxd, 0, -1):
        sel = np.nonzero(depth == lev)[0]
        np.add.at(size, par[sel], size[sel])
    # offsets among siblings: nodes sorted by (parent, pid); exclusive sums of the sizes inside a family
    key = np.where(hp, par, -1).astype(np.int64)
    o = np.lexsort((order, key))
    ks, ss = key[o], size[o]
    cs = np.cumsum(ss) - ss
    first = np.r_[True, ks[1:] != ks[:-1]]
    fam_base = np.maximum.accumulate(np.where(first, cs, 0))
    sib = np.empty(P, np.int64)
    sib[o] = cs - fam_base
    pre = np.zeros(P, np.int64)
    sel = np.nonzero(depth == 0)[0]
    pre[sel] = sib[sel]
    for lev in range(1, maxd + 1):
        sel = np.nonzero(depth == lev)[0]
        pre[sel] = pre[par[sel]] + 1 + sib[sel]
    return pre, depth, maxd


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
    width = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    n_samp = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
    l2_min = 24
    import_kmerdb_amd()
    import importlib
    S = importlib.import_module("kmerdb_amd.synth")
    t0 = time.time()
    g = S.CladeGenomes(n, 50, L, seed=20260929, device="cpu")
    pat = S.build_patterns(lambda i: S.kmers_of(g.sample(i), 18), n, "cpu")
    par = pat["parent"].numpy().astype(np.int64)
    lp = pat["local_ptr"].numpy()
    ids = pat["local_ids"].numpy()
    l = pat["num_local"].numpy()
    w = pat["num_kmers"].numpy()
    ns = pat["num_samples"].numpy()
    P = par.size
    print("samples %d length %d width %d: %d patterns (built in %.0f s)" % (n, L, width, P, time.time() - t0), flush=True)
    NB = (n + width - 1) // width
    NW = (NB + 63) // 64
    pre, depth, maxd = dfs_order(par)
    # block sets of the full lists, as NW 64-bit words per node
    own = np.zeros((P, NW), np.uint64)
    owner = np.repeat(np.arange(P), l)
    blk = ids // width
    np.bitwise_or.at(own, (owner, blk >> 6), np.uint64(1) << (blk & 63).astype(np.uint64))
    full = own
    for lev in range(1, maxd + 1):
        sel = np.nonzero(depth == lev)[0]
        full[sel] |= full[par[sel]]
    cnt = np.zeros(P, np.int64)
    for k in range(NW):
        x = full[:, k].copy()
        c = np.zeros(P, np.int64)
        while True:
            nz = x != 0
            if not nz.any():
                break
            c += nz
            x &= x - np.uint64(1)
        cnt += c
    emit = (w > 0) & (ns >= 2)
    inv = np.argsort(pre)                       # node at DFS position i
    # ---- narrow stream
    first_blk = np.zeros(P, np.int64)
    for k in range(NW - 1, -1, -1):
        x = full[:, k]
        nz = x != 0
        low = (x & (~x + np.uint64(1)))
        fb = np.zeros(P, np.int64)
        fb[nz] = np.log2(low[nz].astype(np.float64)).astype(np.int64) + 64 * k
        first_blk = np.where(nz, fb, first_blk)
    narrow = emit & (cnt <= 2)
    nar_d = narrow[inv]
    fb_d = first_blk[inv]
    SL = 2048
    n_sl = (P + SL - 1) // SL
    top_share, distinct = [], []
    tot_nar = 0
    tot_top = 0
    for s in range(n_sl):
        m = nar_d[s * SL:(s + 1) * SL]
        b = fb_d[s * SL:(s + 1) * SL][m]
        if b.size == 0:
            continue
        u, c = np.unique(b, return_counts=True)
        distinct.append(u.size)
        tot_nar += b.size
        tot_top += c.max()
    distinct = np.array(distinct)
    print("narrow stream: %d emitting nodes with <= 2 blocks in %d slices of %d; distinct first blocks per slice: mean %.2f, 1: %.1f %%, <= 2: %.1f %%, <= 4: %.1f %%; "
          "first-block records in their slice's most frequent block: %.2f %%" %
          (tot_nar, distinct.size, SL, distinct.mean(), 100.0 * (distinct == 1).mean(), 100.0 * (distinct <= 2).mean(), 100.0 * (distinct <= 4).mean(), 100.0 * tot_top / max(1, tot_nar)), flush=True)
    # ---- wide stream
    wide_all = (cnt >= 3)[inv]
    wide_emit_d = (emit & (cnt >= 3) & (cnt < l2_min))[inv]
    wpos = np.nonzero(wide_all)[0]              # the wide list: every node with three blocks or more, DFS order
    n_wide = wpos.size
    n_b = (n_wide + 63) // 64
    rng = np.random.default_rng(1)
    pick = np.sort(rng.choice(n_b // 4, size=min(n_samp // 4, n_b // 4), replace=False))     # runs of 4 batches
    thr = [64, 48, 32, 24, 16, 12, 8, 4, 2, 1]
    rec_by_thr = np.zeros(len(thr), np.int64)
    steps_by_thr = np.zeros(len(thr), np.int64)
    tot_rec = 0
    Ks = [1, 2, 4, 8, 16, 32]
    run_cov = np.zeros(len(Ks), np.int64)
    run_tot = 0
    tiles_per_batch = []
    for r in pick:
        run_tiles = {}
        for bi in range(4):
            b0 = (r * 4 + bi) * 64
            nodes = inv[wpos[b0:b0 + 64]]
            nodes = nodes[wide_emit_d[wpos[b0:b0 + 64]]]
            tl = []
            for nd in nodes:
                bl = []
                for k in range(NW):
                    x = int(full[nd, k])
                    while x:
                        lb = x & -x
                        bl.append(lb.bit_length() - 1 + 64 * k)
                        x ^= lb
                bl = np.array(bl)
                a, b = np.tril_indices(bl.size)
                tl.append(bl[a] * (bl[a] + 1) // 2 + bl[b])
            if not tl:
                continue
            tl = np.concatenate(tl)
            u, c = np.unique(tl, return_counts=True)
            tiles_per_batch.append(u.size)
            tot_rec += tl.size
            for i, t in enumerate(thr):
                m = c >= t
                rec_by_thr[i] += c[m].sum()
                steps_by_thr[i] += m.sum()
            for uu, cc in zip(u.tolist(), c.tolist()):
                run_tiles[uu] = run_tiles.get(uu, 0) + cc
        if run_tiles:
            cc = np.sort(np.array(list(run_tiles.values())))[::-1]
            run_tot += cc.sum()
            for i, K in enumerate(Ks):
                run_cov[i] += cc[:K].sum()
    print("wide stream: %d nodes with >= 3 blocks (%d emitting below the second level's %d blocks), %d batches, %d sampled: %d records, %.1f distinct tiles per batch" %
          (n_wide, int(wide_emit_d.sum()), l2_min, n_b, 4 * pick.size, tot_rec, np.mean(tiles_per_batch)))
    print("  records in tiles with >= T contributing nodes in their batch of 64 (and MFMA steps that would take, per 64 records):")
    for i, t in enumerate(thr):
        print("    T = %2d: %6.2f %% of the records, %8d tile steps = %.2f steps per 64 records covered" %
              (t, 100.0 * rec_by_thr[i] / max(1, tot_rec), steps_by_thr[i], steps_by_thr[i] / max(1.0, rec_by_thr[i] / 64.0)))
    print("  records of a run of 4 batches (256 wide nodes) covered by its K most frequent tiles:")
    for i, K in enumerate(Ks):
        print("    K = %2d: %6.2f %%" % (K, 100.0 * run_cov[i] / max(1, run_tot)))


if __name__ == "__main__":
    main()
