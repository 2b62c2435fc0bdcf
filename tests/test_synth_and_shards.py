"""CPU tests of the synthetic-database generator (kmer-db_amd/synth.py) and of the multi-GPU
sharding scheme of bench.py: k-mer space split by prefix bucket, one partial matrix per rank,
summed with torch.distributed (gloo here, RCCL on the GPU box)."""
import importlib
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def S(K):
    return importlib.import_module("kmerdb_amd.synth")


def _db_from(S, O, g, pat, k, f, td, name):
    arr = S.to_view_arrays(pat)
    path = os.path.join(td, name)
    S.write_db(path, k, f, [g.name(i) for i in range(g.n_samples)], pat["sample_counts"], arr,
               kmers_count=int(pat["dictionary"].numel()))
    return path, arr


@pytest.mark.parametrize("N,cs,L,k,f", [(48, 12, 12000, 18, 1.0), (40, 8, 20000, 25, 0.1), (24, 24, 4000, 21, 1.0)])
def test_generator_reproduces_reference_build(S, O, N, cs, L, k, f):
    g, pat = S.synth_database(N, cs, L, k=k, fraction=f, seed=7)
    with tempfile.TemporaryDirectory() as td:
        path, arr = _db_from(S, O, g, pat, k, f, td, "s.db")
        a = O.OracleDB(path)
        # k-mers: torch extraction == the oracle's restatement of kmer_extract.h on the FASTA text
        for i in (0, N // 2, N - 1):
            seq = g.fasta(i).split("\n")[1].encode()
            assert np.array_equal(O.sort_unique(O.extract_seq(seq, k, f)), S.kmers_of(g.sample(i), k, f).numpy().view(np.uint64))
        # chains strictly increasing, checksum identity
        for pid in range(0, a.P, max(1, a.P // 200)):
            assert np.all(np.diff(a.decode_chain(pid).astype(np.int64)) > 0)
        m = a.all2all_dense()
        assert int(m.astype(np.uint64).sum()) == a.update_counts()["sum_matrix"]
        # brute force: |K_i ∩ K_j| from the raw k-mer sets
        sets = [S.kmers_of(g.sample(i), k, f).numpy() for i in range(N)]
        for i, j in [(1, 0), (N - 1, 0), (N - 1, N - 2), (N // 2, 3)]:
            assert m[i * (i - 1) // 2 + j] == np.intersect1d(sets[i], sets[j]).size
        if O.have_ref():
            # same k-mers through the REAL reference build: same tree shape and same matrix
            O.write_kmers_bin(os.path.join(td, "k.bin"), k, f, [(g.name(i), sets[i].view(np.uint64)) for i in range(N)])
            O.ref_build(os.path.join(td, "k.bin"), os.path.join(td, "r.db"), 2)
            b = O.OracleDB(os.path.join(td, "r.db"))
            assert a.P == b.P
            key = lambda h: sorted(map(tuple, h[:, [0, 2, 3, 4, 5]].tolist()))     # noqa: E731
            assert key(a.pattern_headers()) == key(b.pattern_headers())
            assert np.array_equal(b.all2all_dense(), m)
            mr, _ = O.ref_all2all(path, os.path.join(td, "m.u32"), 2)
            assert np.array_equal(mr, m)


def test_generator_short_genome_forms_equal_the_loop_forms(S, monkeypatch):
    """Collections of short genomes (the GPU suite's 10 000 - 70 000-sample databases) take two shortcuts in the generator: all k-mer
    windows of a genome as one (n, k) view, and phase A's k-mer sets kept for phase B.  Both are bit-identical with the loop forms
    (which long genomes keep): k-mer sets at several k with and without minhash, genomes shorter than k, and whole pattern sets."""
    g = S.CladeGenomes(120, 30, 2500, seed=3)
    for k in (7, 18, 24, 25, 31):
        for f in (1.0, 0.1):
            for i in (0, 31, 119):
                monkeypatch.setattr(S, "_WINDOWS_AT_ONCE", 1 << 16)
                a = S.kmers_of(g.sample(i), k, f)
                monkeypatch.setattr(S, "_WINDOWS_AT_ONCE", 0)
                assert torch.equal(a, S.kmers_of(g.sample(i), k, f)), (k, f, i)
    for L in (0, 5, 17, 18, 19):
        monkeypatch.setattr(S, "_WINDOWS_AT_ONCE", 1 << 16)
        a = S.kmers_of(g.sample(3)[:L], 18)
        monkeypatch.setattr(S, "_WINDOWS_AT_ONCE", 0)
        assert torch.equal(a, S.kmers_of(g.sample(3)[:L], 18)) and (a.numel() == 0) == (L < 18)
    pats = []
    for keep, at_once in ((1 << 26, 1 << 16), (0, 0), (3000, 1 << 16)):          # (kept / loop forms / the cache given up half way)
        monkeypatch.setattr(S, "_KEEP_KMERS", keep)
        monkeypatch.setattr(S, "_WINDOWS_AT_ONCE", at_once)
        pats.append(S.synth_database(300, 30, 600, k=18, seed=11)[1])
    for other in pats[1:]:
        for key, a in pats[0].items():
            b = other[key]
            assert torch.equal(a, b) if isinstance(a, torch.Tensor) else list(a) == list(b), key


def test_prefix_shards_sum_to_full_matrix(S, O):
    N, cs, L, k = 32, 8, 6000, 18
    g = S.CladeGenomes(N, cs, L, seed=11)
    full = S.build_patterns(lambda i: S.kmers_of(g.sample(i), k), N, "cpu")
    with tempfile.TemporaryDirectory() as td:
        pf, _ = _db_from(S, O, g, full, k, 1.0, td, "f.db")
        ref = O.OracleDB(pf).all2all_dense()
        for world in (2, 3):
            acc = np.zeros_like(ref)
            for r in range(world):
                def km(i, r=r, world=world):
                    x = S.kmers_of(g.sample(i), k)
                    return x[((x >> 32) % world) == r]
                part = S.build_patterns(km, N, "cpu")
                pp, _ = _db_from(S, O, g, part, k, 1.0, td, "p%d_%d.db" % (world, r))
                acc += O.OracleDB(pp).all2all_dense()
            assert np.array_equal(acc, ref)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out_dir):
    """One rank of the multi-GPU scheme on CPU: shard by prefix bucket, partial matrix, gloo reduce.
    The partial matrix comes from the oracle here (no GPU in this container); on the GPU box the
    same orchestration calls kmdb_all2all_dense_device and reduces with RCCL (bench.py)."""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from _kmerdb_loader import import_kmerdb_amd
    from oracle import oracle as O
    import_kmerdb_amd()
    S = importlib.import_module("kmerdb_amd.synth")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, cs, L, k = 32, 8, 6000, 18
    g = S.CladeGenomes(N, cs, L, seed=11)

    def km(i):
        x = S.kmers_of(g.sample(i), k)
        return x[((x >> 32) % world) == rank]
    part = S.build_patterns(km, N, "cpu")
    arr = S.to_view_arrays(part)
    path = os.path.join(out_dir, "r%d.db" % rank)
    S.write_db(path, k, 1.0, [g.name(i) for i in range(N)], part["sample_counts"], arr)
    m = torch.from_numpy(O.OracleDB(path).all2all_dense().view(np.int32).copy())
    dist.reduce(m, dst=0, op=dist.ReduceOp.SUM)          # int32 sum == uint32 wrap-around sum bitwise
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), m.numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_reduce_matches_single_database(S, O):
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(_rank_main, args=(2, _free_port(), td), nprocs=2, join=True)
        got = np.load(os.path.join(td, "sum.npy"))
        N, cs, L, k = 32, 8, 6000, 18
        g = S.CladeGenomes(N, cs, L, seed=11)
        full = S.build_patterns(lambda i: S.kmers_of(g.sample(i), k), N, "cpu")
        pf, _ = _db_from(S, O, g, full, k, 1.0, td, "f.db")
        assert np.array_equal(got, O.OracleDB(pf).all2all_dense())


def test_gamma_encoder_of_generator_matches_oracle(S, O):
    rng = np.random.default_rng(5)
    P = 200
    l = rng.integers(1, 40, P)
    ids = [np.sort(rng.choice(5000, size=n, replace=False)) for n in l]
    lp = np.zeros(P + 1, dtype=np.int64)
    lp[1:] = np.cumsum(l)
    pat = {"num_local": torch.from_numpy(l.astype(np.int64)), "local_ptr": torch.from_numpy(lp),
           "local_ids": torch.from_numpy(np.concatenate(ids).astype(np.int64))}
    last, nbits, doff, data = S.gamma_encode_patterns(pat)
    data = data.numpy().view(np.uint64)
    for p in range(P):
        assert int(last[p]) == ids[p][-1]
        if l[p] > 1:
            d = O.gamma_decode(data[int(doff[p]): int(doff[p]) + (int(nbits[p]) + 127) // 128 * 2], int(nbits[p]), int(l[p]))
            assert np.array_equal(d.astype(np.int64), np.diff(ids[p]))
        else:
            assert int(nbits[p]) == 0


def test_synthetic_hashtables_serve_new2all(S, O, K):
    """build_hashtables + write_db(tables=...) give a .db the oracle, the front-end reader and (when present)
    the real reference answer new2all queries from; a query equal to database sample i must see row i of the
    all2all matrix and its own k-mer count on the diagonal."""
    N, cs, L, k = 40, 10, 9000, 18
    g, pat = S.synth_database(N, cs, L, k=k, seed=21)
    arr = S.to_view_arrays(pat)
    tables = S.build_hashtables(pat["dictionary"], pat["kmer_pid"], k)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "s.db")
        S.write_db(path, k, 1.0, [g.name(i) for i in range(N)], pat["sample_counts"], arr,
                   kmers_count=int(pat["dictionary"].numel()), tables=tables)
        o = O.OracleDB(path)
        m = o.all2all_dense()
        fresh = S.CladeGenomes(N, cs, L, seed=22)
        qs = [S.kmers_of(g.sample(i), k).numpy().view(np.uint64) for i in (0, 17, N - 1)]
        qs.append(S.kmers_of(fresh.sample(3), k).numpy().view(np.uint64))
        for qi, q in zip((0, 17, N - 1), qs):
            row = o.one2all(q)
            for j in range(N):
                if j == qi:
                    assert row[j] == q.size
                else:
                    a, b = max(qi, j), min(qi, j)
                    assert row[j] == m[a * (a - 1) // 2 + b]
        h = K.HostDB(path)
        v = h.view_arrays()
        assert np.array_equal(v["slots"], tables[1]) and np.array_equal(v["bucket_offset"], tables[0])
        if O.have_ref():
            O.write_kmers_bin(os.path.join(td, "q.bin"), k, 1.0, [("q%d" % i, q) for i, q in enumerate(qs)])
            rows, _ = O.ref_one2all(path, os.path.join(td, "q.bin"), os.path.join(td, "o.u32"), 2)
            assert np.array_equal(rows.reshape(len(qs), N), np.stack([o.one2all(q) for q in qs]))
