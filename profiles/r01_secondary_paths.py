"""Secondary rows of SURVEY §8 (all2all-sp, new2all) measured on the GPU next to the real reference
(oracle/_ref) on the host, on one database: 1000 samples x <length> bp of the bench's clade model.
usage: python profiles/r01_secondary_paths.py [length=300000] [n_queries=64]   -> one JSON object on stdout"""
import importlib
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402
from oracle import oracle as O  # noqa: E402

K = import_kmerdb_amd()
S = importlib.import_module("kmerdb_amd.synth")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N, cs, k = 1000, 50, 18
dev = torch.device("cuda", 0)
g, pat = S.synth_database(N, cs, L, k=k, seed=20260929, device=dev)
arr = S.to_view_arrays(pat)
tables = S.build_hashtables(pat["dictionary"], pat["kmer_pid"], k)
out = {"database": {"samples": N, "genome_length_bp": L, "k": k, "patterns": int(arr["num_kmers"].size),
                    "kmers": int(pat["dictionary"].numel())}}
with tempfile.TemporaryDirectory() as td:
    path = os.path.join(td, "s.db")
    S.write_db(path, k, 1.0, [g.name(i) for i in range(N)], pat["sample_counts"], arr,
               kmers_count=int(pat["dictionary"].numel()), tables=tables)
    d = K.DeviceDB(K.HostDB(path), device=0, with_hashtables=True)
    uc = O.OracleDB(path, skip_hashtables=True).update_counts()
    # ---- all2all-sp
    sp = d.all2all_sparse()
    t0 = time.perf_counter(); sp = d.all2all_sparse(); wall = time.perf_counter() - t0
    st = d.stats()
    best = None
    for thr in (16, 32):
        txt, info = O.ref_all2all_sp(path, os.path.join(td, "sp.txt"), threads=thr)
        if best is None or info["seconds"] < best[1]["seconds"]:
            best = (txt, info)
    lines = best[0].split(b"\n")
    for i in (1, N // 2, N - 1):
        c, v = sp.row(i)
        assert "".join("%d:%d," % (a + 1, b) for a, b in zip(c, v)).encode() == lines[i]
    out["all2all_sp"] = {"gpu_device_ms": st["kernel_ms"], "gpu_wall_ms_incl_csr_copy": wall * 1e3, "nnz": int(sp.nnz),
                         "gpu_pair_comparisons_per_s": uc["sum_matrix"] / (st["kernel_ms"] * 1e-3),
                         "reference_seconds": best[1]["seconds"], "reference_threads": best[1]["threads"],
                         "reference_pair_comparisons_per_s": uc["sum_matrix"] / best[1]["seconds"], "rows_checked_identical": 3}
    # ---- new2all: fresh strains of the database's clades
    g_more = S.CladeGenomes(N + NQ, cs, L, seed=20260929, device=dev)
    qs = [S.kmers_of(g_more.sample(N + i), k).cpu().numpy().view(np.uint64) for i in range(NQ)]
    got = d.new2all(qs)
    t0 = time.perf_counter(); got = d.new2all(qs); wall = time.perf_counter() - t0
    st = d.stats()
    O.write_kmers_bin(os.path.join(td, "q.bin"), k, 1.0, [("q%d" % i, q) for i, q in enumerate(qs[:8])])
    rows, info = O.ref_one2all(path, os.path.join(td, "q.bin"), os.path.join(td, "o.u32"), 1)
    assert np.array_equal(rows.reshape(8, N), got[:8])
    out["new2all"] = {"queries": NQ, "kmers_per_query": int(np.mean([q.size for q in qs])),
                      "gpu_device_ms_total": st["kernel_ms"], "gpu_wall_ms_total_incl_h2d_d2h": wall * 1e3,
                      "gpu_queries_per_s": NQ / (st["kernel_ms"] * 1e-3),
                      "reference_one2all_seconds_per_query_1_thread": info["seconds"] / 8,
                      "reference_queries_per_s_per_thread": 8 / info["seconds"], "rows_checked_identical": 8}
    # ---- new2all from sequence text: host loader (kmdbh_extract_kmers + kmdbh_sort_unique, one thread) + kmdb_new2all_batch
    #      against kmdb_new2all_batch_seq (extraction, sort and unique on the device)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    texts = [acgt[g_more.sample(N + i).cpu().numpy()].tobytes() for i in range(NQ)]
    t0 = time.perf_counter()
    host_q = [K.sort_unique(K.extract_kmers(t, k)) for t in texts]
    t_host_loader = time.perf_counter() - t0
    assert all(np.array_equal(a, b) for a, b in zip(host_q, qs))
    got2, cnt = d.new2all_seq(texts)
    t0 = time.perf_counter(); got2, cnt = d.new2all_seq(texts); wall_seq = time.perf_counter() - t0
    assert np.array_equal(got2, got) and [int(c) for c in cnt] == [q.size for q in qs]
    out["new2all_from_sequences"] = {"queries": NQ, "bases_per_query": len(texts[0]),
                                     "host_loader_seconds_total_1_thread": t_host_loader,
                                     "gpu_wall_ms_host_loader_path": t_host_loader * 1e3 + wall * 1e3,
                                     "gpu_wall_ms_device_loader_path": wall_seq * 1e3,
                                     "queries_per_s_device_loader_wall": NQ / wall_seq, "rows_checked_identical": NQ}
# ---- db2db (all2all-parts cell): the collection split into two databases of 500 samples (even / odd sample ids)
def part_db(ids, path):
    pat_p = S.build_patterns(lambda i: S.kmers_of(g.sample(ids[i]), k), len(ids), dev)
    arr_p = S.to_view_arrays(pat_p)
    tab_p = S.build_hashtables(pat_p["dictionary"], pat_p["kmer_pid"], k)
    S.write_db(path, k, 1.0, [g.name(i) for i in ids], pat_p["sample_counts"], arr_p, kmers_count=int(pat_p["dictionary"].numel()), tables=tab_p)
    return int(arr_p["num_kmers"].size)


with tempfile.TemporaryDirectory() as td:
    pa, pb = os.path.join(td, "a.db"), os.path.join(td, "b.db")
    n_pa = part_db(list(range(0, N, 2)), pa)
    n_pb = part_db(list(range(1, N, 2)), pb)
    da = K.DeviceDB(K.HostDB(pa), device=0, with_hashtables=True)
    db_ = K.DeviceDB(K.HostDB(pb), device=0, with_hashtables=True)
    got = db_.db2db(da)
    t0 = time.perf_counter(); got = db_.db2db(da); wall = time.perf_counter() - t0
    st = db_.stats()
    best = None
    for thr in (16, 32):
        txt, info = O.ref_db2db_sp(pb, pa, os.path.join(td, "o.txt"), threads=thr)
        if best is None or info["seconds"] < best[1]["seconds"]:
            best = (txt, info)
    lines = best[0].split(b"\n")
    for r in (0, 123, 499):
        assert "".join("%d:%d," % (c + 1, v) for c, v in enumerate(got[r]) if v).encode() == lines[r]
    out["db2db"] = {"rows": int(got.shape[0]), "cols": int(got.shape[1]), "patterns": [n_pb, n_pa], "shared_kmer_pairs": int(got.astype(np.uint64).sum()),
                    "gpu_device_ms": st["kernel_ms"], "gpu_wall_ms_incl_d2h": wall * 1e3,
                    "reference_seconds": best[1]["seconds"], "reference_threads": best[1]["threads"], "rows_checked_identical": 3}
print(json.dumps(out, indent=1))
