#!/usr/bin/env python3
"""bench.py — all2all common-k-mer counting on MI355X: throughput AND wall-clock.

One "step" = one dense all2all call (SimilarityCalculator::all2all's job, reference
src/similarity_calculator.cpp:42-438, called once per run at console_all2all.cpp:31-36) on a synthetic
clade-mutation database resident in HBM.  The call is self-contained: gamma decode, block placement, record
emission and accumulation all happen inside it; upload only converts the on-disk format (DFS order, packed streams).
Reported:
  value / ms_per_step   warm calls (the timed K steps)
  wall.upload_s         kmdb_db_upload: host conversion + H2D + device layout   (the reference's deserialize)
  wall.cold_call_ms     first call after upload, host matrix out (H2D/D2H inclusive)
  wall.cold_total_s     upload + first call = what one `all2all` run costs end to end, next to cpu_baseline.seconds
Workloads (--workload): c2 = BASELINE.json configs[1], 1000 x 5 Mbp; c3part = the sample count of configs[2] on one GPU,
10 000 samples x 300 kbp (rides along in the default line); c3gpu = one GPU's share of configs[2], 10 000 samples x 625 kbp
(5 Mbp / 8 GPUs; more than 2^31 local ids in the pattern tree: the generator places them without a global sort).  With --gpus N the k-mer space is sharded by prefix bucket (kmer >> 32, reference
src/types.h:25-27): --scaling weak (default) keeps per-GPU work fixed (genomes N x longer, rank r owns the buckets
congruent to r mod N); --scaling strong shards ONE database of the workload's size (kmdb_db_upload_shard).  The
partial matrices are summed with one RCCL reduce.  `python bench.py --gpus N` starts its own N ranks.

Prints ONE JSON line on stdout (rank 0); progress goes to stderr.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from _kmerdb_loader import import_kmerdb_amd  # noqa: E402

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s

WORKLOADS = {
    "c2": dict(samples=1000, clade_size=50, length=5_000_000),
    "c3part": dict(samples=10000, clade_size=50, length=300_000),
    "c3gpu": dict(samples=10000, clade_size=50, length=625_000),           # configs[2] / 8 GPUs
    # secondary modes (--mode), sized so that the synthetic generator and the reference finish in minutes
    "c4part": dict(samples=20000, clade_size=50, length=100_000, k=25, fraction=0.1),      # configs[3] is 50 000 samples: --samples 50000
    # all2all-sp on data that IS sparse: the clades descend from independent roots (r1 = 0.75: a clade ancestor shares no 25-mer with another),
    # so only the 50 x 50 blocks on the diagonal are non-zero — configs[3]'s sample count, nnz = 1.2 M of 1.25 G cells
    "c4sparse": dict(samples=50000, clade_size=50, length=50_000, k=25, fraction=0.1, r1=0.75),
    "c5part": dict(samples=10000, clade_size=50, length=100_000, queries=1000),            # configs[4]: 1000 queries vs a 10 000-sample database
    # configs[4] at one GPU's share of configs[2]'s database: 1000 queries vs 10 000 x 625 kbp (hashtables of 140 M k-mers: beyond the caches)
    "c5gpu": dict(samples=10000, clade_size=50, length=625_000, queries=1000),
    "parts": dict(samples=2000, clade_size=50, length=300_000),                            # all2all-parts cell: two halves of one collection
}
MODE_WORKLOAD = {"all2all": "c2", "all2all-sp": "c4part", "new2all": "c5part", "db2db": "parts"}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


ALL2ALL_SOURCES = ("a2a_blocks.hip", "a2a_v1.hip", "device_common.h", "engine.hip", "engine_internal.h", "engine_state.h", "layout.hip", "prim.h")


def sources_sha16():
    """hash of the sources of the all2all call (kmer-db_amd/csrc: pipeline, layout, entry points, shared headers): PMC traffic measured by
    profiles/collect_counters.sh is stamped with it and only replayed into a bench line when that code is still the same (the GPU box
    has no .git to ask for a commit)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "kmer-db_amd", "csrc")
    for fn in ALL2ALL_SOURCES:
        h.update(fn.encode())
        with open(os.path.join(d, fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def replayed_traffic(workload):
    """(bytes per call, source text) from profiles/latest_traffic[_<workload>].json if it was measured on THIS code, else (None, why)"""
    tpath = os.path.join(ROOT, "profiles", "latest_traffic.json" if workload == "c2" else "latest_traffic_%s.json" % workload)
    if not os.path.exists(tpath):
        return None, None
    with open(tpath) as f:
        tj = json.load(f)
    if tj.get("sources_sha16") != sources_sha16():
        return None, "profiles/%s was measured on other sources (%s, this code is %s): not replayed" % (os.path.basename(tpath), tj.get("sources_sha16"), sources_sha16())
    return tj["traffic_bytes_per_pass"], "REPLAYED from profiles/%s (measured on these sources, sha16 %s, not in this run): %s" % (
        os.path.basename(tpath), tj["sources_sha16"], tj["source"])


def build_db(K, S, n_samples, clade_size, length, k, seed, device, rank, world, progress=None, with_items=False, fraction=1.0):
    """patterns of the k-mers whose prefix bucket is owned by `rank` (world == 1: all of them)"""
    g = S.CladeGenomes(n_samples, clade_size, length, seed=seed, device=device)

    def kmers(i):
        return S.kmers_of(g.sample(i), k, fraction, prefix_shard=(rank, world))
    t0 = time.time()
    pat = S.build_patterns(kmers, n_samples, device, progress=progress)
    arr = S.to_view_arrays(pat)
    items = S.shard_item_lists(pat["dictionary"], pat["kmer_pid"], k) if with_items else None
    log("[rank %d] synth db: %d samples x %d bp, %d k-mers, %d patterns in %.1f s" % (
        rank, n_samples, length, pat["dictionary"].numel(), arr["num_kmers"].size, time.time() - t0))
    names = [g.name(i) for i in range(n_samples)]
    return arr, names, pat["sample_counts"], int(pat["dictionary"].numel()), items


def generate_in_child(dev_index, tmp=None, **spec):
    """build_db(**spec) in a process of its own; the arrays come back through files in shared memory.  The synthetic generator
    works through ~100 GB of torch allocations on the GPU; a process that has just handed that much VRAM back to the driver
    twice saw its next kmdb_db_upload take 5.8 s instead of 0.45 s (rounds 1-2, about 2 of 27 runs).  That is the generator's
    cost, and a user's process (the front-end reading a .db) never has it: the timed process now starts its upload on a device
    it has not churned, the way the front-end does, instead of sleeping after torch.cuda.empty_cache()."""
    import shutil
    bases = [tmp] if tmp is not None else []
    if tmp is None:
        try:
            if shutil.disk_usage("/dev/shm").free > (32 << 30):
                bases.append("/dev/shm")
        except OSError:
            pass
        bases.append(tempfile.gettempdir())          # (also the second try when shared memory fills up: several ranks write at once)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                            "TORCHELASTIC_RUN_ID", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE")}
    last = None
    for base in bases:
        td = tempfile.mkdtemp(prefix="kmdb_gen_", dir=base)
        try:
            t0 = time.time()
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--generate-spec", json.dumps(dict(spec, out=td, dev_index=dev_index))], env=env)
            arr = {nm[:-4]: np.load(os.path.join(td, nm)) for nm in sorted(os.listdir(td)) if nm.endswith(".npy") and not nm.startswith("_")}
            with open(os.path.join(td, "meta.json")) as f:
                meta = json.load(f)
            items = None
            if meta["with_items"]:
                items = (np.load(os.path.join(td, "_bucket_offset.npy")), np.load(os.path.join(td, "_items.npy")))
            log("[rank %d] generator process done, arrays read back in %.1f s total" % (spec["rank"], time.time() - t0))
            return arr, meta["names"], meta["sample_counts"], meta["n_kmers"], items
        except (subprocess.CalledProcessError, OSError) as e:
            last = e
            log("[rank %d] generator process with its files under %s failed (%s)" % (spec["rank"], base, e))
        finally:
            shutil.rmtree(td, ignore_errors=True)
    raise SystemExit("bench.py: the synthetic generator failed: %s" % last)


def generator_child(spec):
    """the other side of generate_in_child()"""
    torch.cuda.set_device(spec["dev_index"])
    device = torch.device("cuda", spec["dev_index"])
    K = import_kmerdb_amd()
    import importlib
    S = importlib.import_module("kmerdb_amd.synth")
    arr, names, counts, nk, items = build_db(K, S, spec["n_samples"], spec["clade_size"], spec["length"], spec["k"], spec["seed"], device, spec["rank"], spec["world"],
                                             progress=spec.get("progress"), with_items=spec.get("with_items", False), fraction=spec.get("fraction", 1.0))
    for nm, a in arr.items():
        np.save(os.path.join(spec["out"], nm + ".npy"), a)
    if items is not None:
        np.save(os.path.join(spec["out"], "_bucket_offset.npy"), items[0])
        np.save(os.path.join(spec["out"], "_items.npy"), items[1])
    with open(os.path.join(spec["out"], "meta.json"), "w") as f:
        json.dump({"names": names, "sample_counts": [int(c) for c in counts], "n_kmers": int(nk), "with_items": items is not None}, f)


def release_generator_memory(rank):
    """torch's cached VRAM goes back to the driver before an upload (the big generator runs in a process of its own, see
    generate_in_child(); what is left in this process are the small tensors of earlier steps)"""
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    log("[rank %d] device memory before the upload: %.0f of %.0f GB free" % (rank, free / 1e9, total / 1e9))


def upload(K, arr, n_samples, k, device_index, items=None, prefix_shard=None):
    view = K.make_view(k, n_samples, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"],
                       arr["last_sample_id"], arr["num_bits"], arr["data_offset"], arr["data"],
                       bucket_offset=None if items is None else items[0], slots=None if items is None else items[1])
    had = os.environ.get("KMDB_VERBOSE")
    os.environ["KMDB_VERBOSE"] = "1"                      # the phases of the upload go to stderr (a slow one shows where)
    t0 = time.perf_counter()
    d = K.DeviceDB(view, device=device_index, prefix_shard=prefix_shard)
    dt = time.perf_counter() - t0
    if had is None:
        os.environ.pop("KMDB_VERBOSE", None)
    return d, dt


def cpu_baseline(K, S, O, args, device, arr, names, counts, nk, gpu_matrix):
    """The real reference hot path (oracle/_ref) on the host cores, on the FULL database of the timed workload:
    (threads, bufferMb) picked by a sweep on a 1/50-length sample, then one run on the full .db; the interval is the
    reference's own "Calculating matrix of common k-mers" (console_all2all.cpp:31-36).  Without oracle/_ref: the
    oracle's C restatement on the sample only (kind "port")."""
    cores = os.cpu_count() or 1
    L = args.cpu_sample_length
    sarr, snames, scounts, snk, _ = build_db(K, S, args.samples, args.clade_size, L, args.k, args.seed, device, 0, 1)
    with tempfile.TemporaryDirectory(dir=args.tmp) as td:
        spath = os.path.join(td, "sample.db")
        S.write_db(spath, args.k, 1.0, snames, scounts, sarr, kmers_count=snk)
        odb = O.OracleDB(spath, skip_hashtables=True)
        d, _ = upload(K, sarr, args.samples, args.k, device.index or 0)
        gpu_s = d.all2all_dense()
        d.close()
        if not O.have_ref():
            t0 = time.time()
            m = odb.all2all_dense()
            secs = time.time() - t0
            assert np.array_equal(m, gpu_s), "GPU result differs from the CPU baseline on the sample database"
            uc = odb.update_counts()
            return {"value": uc["sum_matrix"] / secs, "unit": "kmer-pair-comparisons/s", "cores": 1, "host_cores": cores, "kind": "port",
                    "seconds": secs, "sample": "oracle C restatement on the same %d-sample model at genome length %d bp (oracle/_ref not built)"
                    % (args.samples, L)}
        # the sample: the reference's matrix == the GPU's on a second, small database of the same model
        m, info = O.ref_all2all(spath, os.path.join(td, "m.u32"), threads=min(cores, 16), buffer_mb=8)
        assert np.array_equal(m, gpu_s), "GPU result differs from the reference on the sample database"
        # the full database: written once in the reference's format.  The reference's 4-stage pipeline does not scale to hundreds of
        # threads (its README.md:185 recommends tuning -t / -buffer): -t and -buffer are swept AT FULL SIZE (the choice
        # used to come from a 1/50-length sample and flipped between rounds) and the best run is the baseline
        t0 = time.time()
        path = os.path.join(td, "full.db")
        S.write_db_fast(path, args.k, 1.0, names, counts, arr, kmers_count=nk, device=device)
        log("  full .db written: %.1f GB in %.1f s" % (os.path.getsize(path) / 1e9, time.time() - t0))
        best, tried = None, []
        # (rounds 4 - 5 swept -t 16 / 32 / 64 / 128: 64 and 128 threads lost every time — 19.6 and 56.3 s against 12.5 at C2 — and cost 76 s of the
        # default run, which now spends them on the secondary rows; --cpu-sweep full brings the five points back)
        t_list = (16, 32, 64, 128) if args.cpu_sweep == "full" else (16, 32)
        for thr, buf in [(t, 8) for t in sorted({min(cores, t) for t in t_list})] + [(None, 32)]:
            if thr is None:
                thr = best[2]                                    # -buffer at the best thread count
            t0 = time.time()
            m, info = O.ref_all2all(path, os.path.join(td, "full.u32"), threads=thr, buffer_mb=buf)
            proc = time.time() - t0
            tried.append((thr, buf, round(info["seconds"], 2)))
            assert np.array_equal(m, gpu_matrix), "GPU matrix differs from the reference's on the full database"
            if best is None or info["seconds"] < best[1]["seconds"]:
                best = (m, info, thr, buf, proc)
        log("  reference on the full database, (threads, bufferMb, compute s):", tried)
        m, info, thr, buf, ref_process_s = best
        sum_pairs = float(m.astype(np.uint64).sum())
        fe = frontend_run(K, path, td, m, names, counts, args.k)
        log("  front-end on the same .db: %.2f s whole process (compute %.3f s, CSV %.3f s), CSV == the reference matrix's" % (
            fe["frontend_s"], fe["frontend_compute_s"] or -1, fe["frontend_csv_s"] or -1))
        return {"value": sum_pairs / info["seconds"], "unit": "kmer-pair-comparisons/s", "cores": thr, "host_cores": cores, "kind": "reference",
                "seconds": info["seconds"], "process_seconds": ref_process_s, "buffer_mb": buf, "frontend": fe,
                "sweep": tried,
                "sample": "full %s database (%d patterns), reference SimilarityCalculator::all2all compute interval, -t %d -buffer %d "
                          "(best of a %d-point sweep of -t / -buffer on the FULL database: `sweep`); the whole %d-cell GPU matrix compared equal in every run"
                          % (args.workload, arr["num_kmers"].size, thr, buf, len(tried), m.size)}


def frontend_run(K, db_path, td, ref_matrix, names, counts, k):
    """`kmer-db-amd all2all <db> <csv>` — the product's front-end as a user runs it: .db read from disk, upload, the call, CSV written —
    timed as a whole process next to the reference's (console_all2all.cpp:25-78); the CSV compared byte for byte with one formatted
    from the REFERENCE's raw matrix (same row format: array.h:254-257, conversion.h:99-165)."""
    exe = os.path.join(ROOT, "kmer-db_amd", "bin", "kmer-db-amd")
    out_csv = os.path.join(td, "frontend.csv")
    t0 = time.time()
    r = subprocess.run([exe, "all2all", db_path, out_csv], capture_output=True, text=True, env=dict(os.environ, KMDB_VERBOSE="1"))
    wall = time.time() - t0
    log("front-end:\n  " + "\n  ".join(ln for ln in r.stderr.splitlines() if ln.startswith(("[kmdb] main", "[kmdb] upload part", "[kmdb] load", "Database", "OK", "Process"))))
    if r.returncode != 0:
        raise SystemExit("bench.py: the front-end failed: %s" % r.stderr[-2000:])
    import re
    secs = [float(x) for x in re.findall(r"OK \(([0-9.eE+-]+) seconds\)", r.stderr)]
    lu = re.search(r"Database loaded in ([0-9.eE+-]+) s, uploaded in ([0-9.eE+-]+) s", r.stderr)
    up = re.search(r"Process up for ([0-9.eE+-]+) s", r.stderr)
    res = {"frontend_s": wall, "frontend_load_s": float(lu.group(1)) if lu else None, "frontend_upload_s": float(lu.group(2)) if lu else None,
           "frontend_compute_s": secs[0] if secs else None, "frontend_csv_s": secs[1] if len(secs) > 1 else None,
           "frontend_process_up_s": float(up.group(1)) if up else None,      # from the kernel's start of the process to the table on disk; the rest of frontend_s is the process' end
           "frontend_csv_bytes": os.path.getsize(out_csv)}
    with open(out_csv, "rb") as f:
        got = f.read()
    # the same command with the ordinary process teardown (KMDB_FULL_TEARDOWN=1: host image and device pools freed one by one, runtime exit
    # handlers): what ending the process at once is worth
    t0 = time.time()
    r2 = subprocess.run([exe, "all2all", db_path, out_csv], capture_output=True, text=True,
                        env=dict(os.environ, KMDB_FULL_TEARDOWN="1", KMDB_LOAD_POPULATE="0", KMDB_VERBOSE="1"))
    res["frontend_full_teardown_s"] = time.time() - t0
    log("front-end with the ordinary teardown and without the bulk prefault of the reader's arrays:\n  " +
        "\n  ".join(ln for ln in r2.stderr.splitlines() if ln.startswith(("[kmdb] load", "Database", "Process"))))
    with open(out_csv, "rb") as f:
        assert r2.returncode == 0 and f.read() == got, "front-end with full teardown: different output"
    # expected text: header lines + one row per sample from the reference's matrix
    n = len(names)
    h = K.HostDB(db_path, skip_hashtables=True)
    exp = [K.format_header(h)]
    h.close()
    m = np.ascontiguousarray(ref_matrix, np.uint32)
    for i in range(n):
        exp.append(K.format_dense_row(names[i], int(counts[i]), m[i * (i - 1) // 2: i * (i - 1) // 2 + i]))
    exp = b"".join(exp)
    assert got == exp, "front-end CSV differs from the CSV of the reference's matrix"
    res["frontend_csv_matches_reference_matrix"] = True
    os.unlink(out_csv)
    return res


def rows_from_definition(S, M, wl, k, seed, device, rows):
    """Rows of the matrix M (device, lower triangle) recomputed straight from the definition — M[i][j] = |K_i ∩ K_j| over the samples'
    k-mer SETS, no pattern, tree or record involved — for a database too large for the reference's host image: the genomes are
    derived again from the seed, the k-mer sets of the checked rows kept sorted, every other sample's set looked up in them."""
    N = wl["samples"]
    g = S.CladeGenomes(N, wl["clade_size"], wl["length"], seed=seed, device=device)
    rows = sorted(set(int(r) for r in rows if 0 < r < N))
    sets = [S.kmers_of(g.sample(i), k) for i in rows]
    want = [torch.zeros(i, dtype=torch.int64, device=device) for i in rows]
    top = max(rows)
    for j in range(top):
        kj = S.kmers_of(g.sample(j), k)
        for t, i in enumerate(rows):
            if j >= i:
                continue
            si = sets[t]
            pos = torch.searchsorted(si, kj).clamp_(max=si.numel() - 1)
            want[t][j] = (si[pos] == kj).sum()
    bad = []
    for t, i in enumerate(rows):
        o = i * (i - 1) // 2
        if not torch.equal(M[o: o + i].to(torch.int64).bitwise_and(0xFFFFFFFF), want[t]):
            bad.append(i)
    return rows, bad


def extra_workload(K, S, args, device, name, reference=True, definition_rows=0):
    """The default run also times the 10 000-sample workloads and embeds them in the one JSON line: c3part (BASELINE configs[2]'s sample
    count at a genome length whose database the reference can hold: the whole matrix compared with the real reference's, oracle/_ref, on
    the same database written in the reference's format) and c3gpu (one GPU's share of configs[2] itself, 10 000 x 625 kbp: too large
    for the reference's host image here, so `definition_rows` rows are recomputed from the samples' k-mer sets instead).  Warm calls
    (HIP events around the whole call) and the checksum identity for both."""
    wl = WORKLOADS[name]
    dev_index = device.index or 0
    arr, names, counts, nk, _ = generate_in_child(dev_index, n_samples=wl["samples"], clade_size=wl["clade_size"], length=wl["length"], k=args.k, seed=args.seed,
                                                  rank=0, world=1)
    release_generator_memory(0)
    db, upload_s = upload(K, arr, wl["samples"], args.k, dev_index)
    t0 = time.perf_counter()
    first = db.all2all_dense()
    cold_ms = (time.perf_counter() - t0) * 1e3
    st0 = db.stats()
    cells = db.tri_size()
    M = torch.zeros(max(cells, 1), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(args.warmup):
        db.all2all_dense_device(M.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    ms, parts = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        db.all2all_dense_device(M.data_ptr(), stream=stream)
        _s = db.stats()
        ms.append(_s["kernel_ms"])
        parts.append((_s["k0_ms"], _s["k1n_ms"], _s["k1g_ms"], _s["k2_ms"]))
        assert _s["sized_call"] == 0, "%s: a warm call measured its launch sizes again" % name
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) / args.steps * 1e3
    got = int(M[:cells].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())
    assert got == int(st0["sum_pairs"]), "%s: matrix checksum mismatch" % name
    assert np.array_equal(M[:cells].cpu().numpy().view(np.uint32), first), "%s: warm call differs from the first call" % name
    stl = db.stats()
    kern_ms = float(np.mean(ms))
    pk = np.mean(np.array(parts), axis=0)
    alg = st0["algorithmic_bytes"]
    traffic, traffic_src = replayed_traffic(name)
    out = {"workload": "%s: %d synthetic %g Mbp genomes (clades of %d), k=%d f=1.0, dense all2all" % (name, wl["samples"], wl["length"] / 1e6, wl["clade_size"], args.k),
           "ms_per_step": wall_ms, "kernel_ms": kern_ms, "step_kernel_ms": [round(float(x), 3) for x in ms],
           "value": float(st0["sum_pairs"]) / (wall_ms * 1e-3), "unit": "kmer-pair-comparisons/s",
           "roofline": {"bound": "hbm", "achieved": alg / (kern_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg},
           "per_kernel_ms": {"decode": float(pk[0]), "emit_narrow": float(pk[1]), "wide_list+emit_wide": float(pk[2]), "apply": float(pk[3])},
           "patterns": db.P, "records": stl["n_records"], "records_applied_from_slices": stl["n_direct"], "wide_nodes": stl["n_wide"], "nodes_joined_per_tile": stl["n_joined"], "record_chunks": stl["n_chunks"], "block_width": stl["width"],
           "path": {1: "block-record pipeline", 2: "v1 tile kernel", 3: "v1 HBM-atomics kernel"}.get(stl["path"], "none"),
           "upload_s": upload_s, "cold_call_ms": cold_ms, "checks": "sum of the matrix == sum_p w_p C(n_p, 2); warm == cold", "reference_match": None}
    db.close()
    out["rows_from_definition"] = False
    if definition_rows:
        N, cs = wl["samples"], wl["clade_size"]
        cand = [1, 2, cs - 1, cs, cs + 1, N // 2, N // 2 + 1, N - cs, N - cs - 1, N - 2, N - 1, N // 3, 2 * N // 3][:max(8, definition_rows)]
        t1 = time.time()
        rows, bad = rows_from_definition(S, M, wl, args.k, args.seed, device, cand)
        assert not bad, "%s: rows %s differ from |K_i ∩ K_j| over the samples' k-mer sets" % (name, bad)
        out["rows_from_definition"] = True
        out["definition_rows"] = rows
        out["checks"] += "; rows %s == |K_i ∩ K_j| recomputed from the samples' k-mer sets (%.0f s)" % (rows, time.time() - t1)
    del M
    if not reference:
        out["reference_note"] = ("the reference is not run on this database: its host image (patterns + hashtables of %d patterns) and its %d-thread buffers "
                                 "need more host RAM than a shared GPU box guarantees; c3part carries the whole-matrix comparison at the same sample count" % (out["patterns"], 16))
    if reference and not args.no_cpu_baseline:
        from oracle import oracle as O
        if O.have_ref():
            with tempfile.TemporaryDirectory(dir=args.tmp) as td:
                path = os.path.join(td, "full.db")
                S.write_db_fast(path, args.k, 1.0, names, counts, arr, kmers_count=nk, device=device)
                m, info = O.ref_all2all(path, os.path.join(td, "full.u32"), threads=min(os.cpu_count() or 1, 16), buffer_mb=8)
                assert np.array_equal(m, first), "%s: GPU matrix differs from the reference's" % name
                out["reference_match"] = True
                out["reference_compute_s"] = info["seconds"]
                out.update(frontend_run(K, path, td, m, names, counts, args.k))
                out["checks"] += "; the whole %d-cell matrix == the real reference's (all2all, -t %d -buffer 8) on the same database" % (m.size, min(os.cpu_count() or 1, 16))
    return out


def measured_copy_gbs(device, gib=2):
    """achievable HBM bandwidth on this box: a device-to-device copy of `gib` GiB (read + write counted), best of 5"""
    n = gib << 28
    a = torch.empty(n, dtype=torch.int32, device=device)
    b = torch.ones(n, dtype=torch.int32, device=device)
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        a.copy_(b)
        e1.record()
        e1.synchronize()
        best = max(best, 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    torch.cuda.empty_cache()
    return best


def pattern_bytes(arr):
    """B_pat of SURVEY 8(d): the on-disk pattern section, 40 B header + 16 B per 128 stream bits"""
    return int((40 + 16 * ((arr["num_bits"].astype(np.int64) + 127) // 128)).sum())


def secondary_mode(args, K, S, device, embedded=False):
    """--mode all2all-sp | new2all | db2db: the rows of SURVEY 8 beside the dense all2all, one JSON line each in the same contract.
    `value` is on device time (HIP events around the call's kernels, kmdb_stats.kernel_ms); these entry points take and return
    HOST buffers, so the PCIe-inclusive wall time of one call is reported beside it (wall.call_ms)."""
    from oracle import oracle as O
    dev_index = device.index or 0
    N, cs, L, k, f = args.samples, args.clade_size, args.length, args.k, args.fraction
    g = S.CladeGenomes(N, cs, L, r1=args.r1, seed=args.seed, device=device)
    t0 = time.time()

    def make(ids, with_tables):
        pat = S.build_patterns(lambda i: S.kmers_of(g.sample(ids[i]), k, f), len(ids), device, progress=None)
        arr = S.to_view_arrays(pat)
        tables = S.build_hashtables(pat["dictionary"], pat["kmer_pid"], k) if with_tables else None
        return pat, arr, tables

    def up(arr, n, tables):
        view = K.make_view(k, n, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"], arr["last_sample_id"], arr["num_bits"],
                           arr["data_offset"], arr["data"], bucket_offset=None if tables is None else tables[0],
                           slots=None if tables is None else tables[1])
        t1 = time.perf_counter()
        d = K.DeviceDB(view, device=dev_index, with_hashtables=tables is not None)
        return d, time.perf_counter() - t1

    def timed(fn):
        for _ in range(args.warmup):
            fn()
        dev_ms, t1 = [], time.perf_counter()
        for _ in range(args.steps):
            res, d = fn()
            dev_ms.append(d.stats()["kernel_ms"])
        return res, float(np.mean(dev_ms)), (time.perf_counter() - t1) / args.steps * 1e3

    cores = os.cpu_count() or 1
    desc = "%d synthetic %g Mbp genomes (clade-mutation model, clades of %d), k=%d f=%g" % (N, L / 1e6, cs, k, f)
    with tempfile.TemporaryDirectory(dir=args.tmp) as td:
        if args.mode == "all2all-sp":
            pat, arr, _ = make(list(range(N)), False)
            log("synth db: %d k-mers, %d patterns in %.1f s" % (pat["dictionary"].numel(), arr["num_kmers"].size, time.time() - t0))
            d, upload_s = up(arr, N, None)
            sp, dev_ms, call_ms = timed(lambda: (d.all2all_sparse(), d))
            st = d.stats()
            alg = pattern_bytes(arr) + 8 * int(sp.nnz)
            units, metric, unit = float(st["sum_pairs"]), "all2all-sp k-mer pair-comparisons/sec", "kmer-pair-comparisons/s"
            # identity over the whole output: the non-zeros sum to sum_p w_p C(n_p, 2) (mod 2^32 per cell is not hit at these sizes)
            assert int(sp.val.astype(np.uint64).sum()) == int(st["sum_pairs"]), "sparse output checksum mismatch"
            cpu = None
            if not args.no_cpu_baseline and O.have_ref():
                path = os.path.join(td, "db.db")
                S.write_db_fast(path, k, f, [g.name(i) for i in range(N)], pat["sample_counts"], arr, kmers_count=int(pat["dictionary"].numel()), device=device)
                txt, info = O.ref_all2all_sp(path, os.path.join(td, "sp.txt"), threads=min(cores, 16))
                lines = txt.split(b"\n")
                for i in range(N):
                    c, v = sp.row(i)
                    assert "".join("%d:%d," % (a + 1, b) for a, b in zip(c, v)).encode() == lines[i], "row %d differs from the reference" % i
                cpu = {"value": units / info["seconds"], "unit": unit, "cores": min(cores, 16), "host_cores": cores, "kind": "reference", "seconds": info["seconds"],
                       "sample": "the same database, reference all2all_sp compute interval; all %d rows of the sparse output compared equal" % N}
            # the same call with a bound on a measure: cells that miss it never leave the device
            cnt32 = np.asarray(pat["sample_counts"], dtype=np.uint64).astype(np.uint32)
            rows_of = np.repeat(np.arange(N), np.diff(sp.row_ptr).astype(np.int64))
            a_, b_, c_ = cnt32[rows_of].astype(np.uint32), cnt32[sp.col].astype(np.uint32), sp.val.astype(np.uint32)
            jac = c_.astype(np.float64) / (a_ + b_ - c_).astype(np.float64)
            thr = float(np.quantile(jac, 0.99)) if jac.size else 0.0
            t1 = time.perf_counter()
            spf = d.all2all_sparse_filtered([("jaccard", thr, None)], cnt32, measure="jaccard")
            filt_ms = (time.perf_counter() - t1) * 1e3
            keep = jac >= thr
            assert spf.nnz == int(keep.sum()) and np.array_equal(spf.col, sp.col[keep]) and np.array_equal(spf.val, sp.val[keep]) \
                and np.array_equal(spf.measure, jac[keep]), "filtered sparse output differs"
            extra_wall = {"filtered_call_ms": filt_ms, "filtered_nnz": int(spf.nnz), "filter": "jaccard >= %.6g (the 99th percentile), measure = jaccard" % thr}
            cfg = {"workload": "%s: %s, all2all-sp" % (args.workload, desc), "nnz": int(sp.nnz), "cells": int(N) * (int(N) - 1) // 2, "patterns": int(d.P),
                   "r1": args.r1}
            kernel = "kmdb_all2all_sparse: block-record pipeline into the dense triangle + compaction of the tiles it added to (row_tiles_kernel; CSR)"
        elif args.mode == "new2all":
            NQ = args.queries
            pat, arr, tables = make(list(range(N)), True)
            log("synth db + hashtables: %d k-mers, %d patterns in %.1f s" % (pat["dictionary"].numel(), arr["num_kmers"].size, time.time() - t0))
            d, upload_s = up(arr, N, tables)
            # queries = fresh strains of 20 clades of the collection
            n_clades = max(1, N // cs)
            chosen = [int(c) for c in np.random.default_rng(args.seed + 1000).choice(n_clades, size=min(20, n_clades), replace=False)]
            qs_dev = [S.kmers_of(g.strain(chosen[i * len(chosen) // NQ], N + i), k, f) for i in range(NQ)]
            qs = [q.cpu().numpy().view(np.uint64) for q in qs_dev]
            got, dev_ms, call_ms = timed(lambda: (d.new2all(qs), d))
            # algorithmic bytes: 16 B per looked-up k-mer + the root-path bytes of every distinct hit pattern + 4 N per query
            node_b = torch.from_numpy(40 + 16 * ((arr["num_bits"].astype(np.int64) + 127) // 128)).to(device)
            par = torch.from_numpy(arr["parent_id"].astype(np.int64)).to(device)
            path_b, hop = node_b.clone(), par.clone()
            while bool((hop >= 0).any()):
                live = hop >= 0
                path_b[live] += path_b[hop[live]]
                hop[live] = hop[hop[live]]
            dic, kp = pat["dictionary"], pat["kmer_pid"]
            alg, hits_total = 0, 0
            # second numerator: what a kernel that visits every node of the UNION of the hit patterns' root paths once per query has to
            # read (SURVEY's figure charges every hit pattern its whole root path); exact on every 25th query, scaled to all of them
            seen = torch.zeros(par.numel(), dtype=torch.bool, device=device)
            alg_union, alg_union_full, n_union = 0, 0, 0
            for qi, q in enumerate(qs_dev):
                idx = torch.searchsorted(dic, q).clamp_(max=dic.numel() - 1)
                hit = dic[idx] == q
                pids = torch.unique(kp[idx[hit]].to(torch.int64))
                hits_total += int(hit.sum())
                full = 16 * int(q.numel()) + int(path_b[pids].sum()) + 4 * N
                alg += full
                if qi % 25 == 0:
                    seen.zero_()
                    cur, ub = pids, 0
                    while cur.numel():
                        cur = torch.unique(cur[~seen[cur]])
                        seen[cur] = True
                        ub += int(node_b[cur].sum())
                        cur = par[cur]
                        cur = cur[cur >= 0]
                    alg_union += 16 * int(q.numel()) + ub + 4 * N
                    alg_union_full += full
                    n_union += 1
            alg_union_scaled = int(alg * (alg_union / max(1, alg_union_full)))
            # identity over the whole output: row sums == sum over hit k-mers of the number of samples of their pattern
            ns = torch.from_numpy(arr["num_samples"].astype(np.int64)).to(device)
            for i in (0, NQ // 2, NQ - 1):
                idx = torch.searchsorted(dic, qs_dev[i]).clamp_(max=dic.numel() - 1)
                hit = dic[idx] == qs_dev[i]
                assert int(ns[kp[idx[hit]].to(torch.int64)].sum()) == int(got[i].astype(np.uint64).sum()), "row %d checksum mismatch" % i
            units, metric, unit = float(NQ), "new2all queries/sec (k-mer sets resident on the host, %d-sample database)" % N, "queries/s"
            cpu = None
            # rows straight from the definition — |Q ∩ K_j| over the k-mer SETS, no pattern, table or tree involved — for 16 queries, on the
            # queries' own clades and a spread over all samples
            t1 = time.time()
            spread = sorted(set(range(0, N, max(1, N // 400))))
            checked = []
            for qi in range(0, NQ, max(1, NQ // 16)):
                c0 = chosen[qi * len(chosen) // NQ] * cs
                cols = sorted(set(spread) | set(range(c0, min(N, c0 + cs))))
                want = np.array([int(torch.isin(S.kmers_of(g.sample(j), k, f), qs_dev[qi]).sum()) for j in cols], dtype=np.uint32)
                assert np.array_equal(got[qi][cols], want), "new2all row %d differs from the definition" % qi
                checked.append(qi)
            log("  rows of queries %s == |Q ∩ K_j| on %d samples each (%.0f s)" % (checked, len(cols), time.time() - t1))
            big = int(d.P) > 40_000_000
            if big:
                log("  reference not run: writing its .db (patterns + hashtables of %d patterns) and its host image are beyond this run" % int(d.P))
            if not args.no_cpu_baseline and O.have_ref() and not big:
                path = os.path.join(td, "db.db")
                S.write_db(path, k, f, [g.name(i) for i in range(N)], pat["sample_counts"], arr, kmers_count=int(pat["dictionary"].numel()), tables=tables)
                # the reference's new2all the way its console runs it (console_new2all.cpp:64-95): T worker threads over ALL the queries, one
                # one2all<false> per query on the shared database; T swept, wall clock of the whole batch; every row compared
                O.write_kmers_bin(os.path.join(td, "q.bin"), k, f, [("q%d" % i, q) for i, q in enumerate(qs)])
                best, tried = None, []
                for T in sorted({min(cores, t) for t in ((16,) if embedded else (16, 64, 128))}):      # (riding along in the default line: the thread count that won every sweep so far)
                    rows, info = O.ref_new2all(path, os.path.join(td, "q.bin"), os.path.join(td, "o.u32"), T)
                    tried.append((T, round(info["seconds"], 3)))
                    if best is None or info["seconds"] < best[1]["seconds"]:
                        best = (rows, info, T)
                log("  reference new2all, (threads, s):", tried)
                assert np.array_equal(best[0].reshape(NQ, N), got), "new2all rows differ from the reference"
                cpu = {"value": NQ / best[1]["seconds"], "unit": unit, "cores": best[2], "host_cores": cores, "kind": "reference", "seconds": best[1]["seconds"],
                       "sample": "the same database, ALL %d queries through the reference's new2all compute (T worker threads, one one2all<false> per "
                                 "query, src/console_new2all.cpp:64-95; best of T in %s); all rows compared equal" % (NQ, [t for t, _ in tried])}
            cfg = {"workload": "%s: %d fresh strains (%d k-mers each) against %s, new2all dense" % (args.workload, NQ, int(np.mean([q.size for q in qs])), desc),
                   "queries": NQ, "kmers_found": hits_total, "patterns": int(d.P), "rows_from_definition": checked,
                   "hashtable_slots": int(tables[1].size)}
            kernel = "kmdb_new2all_batch: n2a_probe_kernel + pattern climb + row accumulation"
        else:
            ids_a, ids_b = list(range(0, N, 2)), list(range(1, N, 2))
            pa, arr_a, tab_a = make(ids_a, True)
            pb, arr_b, tab_b = make(ids_b, True)
            log("synth parts + hashtables: %d + %d k-mers in %.1f s" % (pa["dictionary"].numel(), pb["dictionary"].numel(), time.time() - t0))
            da, ua = up(arr_a, len(ids_a), tab_a)
            d, ub = up(arr_b, len(ids_b), tab_b)
            upload_s = ua + ub
            got, dev_ms, call_ms = timed(lambda: (d.db2db(da), d))
            # algorithmic bytes: every stored k-mer of the row database looked up in the column database (8 B item read + 8 B item probed),
            # both pattern sections once, the result once
            alg = 16 * int(pb["dictionary"].numel()) + pattern_bytes(arr_a) + pattern_bytes(arr_b) + 4 * len(ids_a) * len(ids_b)
            units, metric, unit = float(got.astype(np.uint64).sum()), "db2db (all2all-parts cell) shared k-mer pair-comparisons/sec", "kmer-pair-comparisons/s"
            # identity over the whole output: the cell of the union matrix — sum == sum over shared k-mers of n_a(p) * n_b(q)
            common = torch.isin(pb["dictionary"], pa["dictionary"])
            ia = torch.searchsorted(pa["dictionary"], pb["dictionary"][common])
            na = torch.from_numpy(arr_a["num_samples"].astype(np.int64)).to(device)[pa["kmer_pid"][ia].to(torch.int64)]
            nb_ = torch.from_numpy(arr_b["num_samples"].astype(np.int64)).to(device)[pb["kmer_pid"][common].to(torch.int64)]
            assert int((na * nb_).sum()) == int(units), "db2db checksum mismatch"
            cpu = None
            if not args.no_cpu_baseline and O.have_ref():
                fa, fb = os.path.join(td, "a.db"), os.path.join(td, "b.db")
                S.write_db(fa, k, f, [g.name(i) for i in ids_a], pa["sample_counts"], arr_a, kmers_count=int(pa["dictionary"].numel()), tables=tab_a)
                S.write_db(fb, k, f, [g.name(i) for i in ids_b], pb["sample_counts"], arr_b, kmers_count=int(pb["dictionary"].numel()), tables=tab_b)
                txt, info = O.ref_db2db_sp(fb, fa, os.path.join(td, "o.txt"), threads=min(cores, 16))
                lines = txt.split(b"\n")
                for r in range(got.shape[0]):
                    assert "".join("%d:%d," % (c + 1, v) for c, v in enumerate(got[r]) if v).encode() == lines[r], "row %d differs from the reference" % r
                cpu = {"value": units / info["seconds"], "unit": unit, "cores": min(cores, 16), "host_cores": cores, "kind": "reference", "seconds": info["seconds"],
                       "sample": "the same two databases, reference db2db_sp compute interval; all %d rows compared equal" % got.shape[0]}
            cfg = {"workload": "%s: %s split into two databases of %d / %d samples (even / odd ids), db2db" % (args.workload, desc, len(ids_b), len(ids_a)),
                   "patterns": [int(d.P), int(da.P)]}
            kernel = "kmdb_db2db_dense: item probe of the row database's tables in the column database's + pair-of-paths accumulation"
    achieved = alg / (dev_ms * 1e-3) / 1e9
    out = {"metric": metric, "value": units / (dev_ms * 1e-3), "unit": unit, "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dev_ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": dict(cfg, samples=N, genome_length_bp=L, k=k, fraction=f, mode=args.mode),
           "wall": {"upload_s": upload_s, "call_ms": call_ms, **(extra_wall if args.mode == "all2all-sp" else {}),
                    "note": "ms_per_step / value: device time of one call (HIP events); call_ms: the same call as the host sees it, host buffers in "
                            "and out (H2D / D2H inclusive)"},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "kernel": kernel, "kernel_ms": dev_ms, "algorithmic_bytes_per_launch": alg}}
    if args.mode == "new2all":
        # both numerators side by side: SURVEY 8d's (every hit pattern's whole root path) and the union of the root paths per query
        # (what the walk kernel actually has to read: shared ancestors once per query)
        out["roofline"]["union_of_root_paths"] = {"algorithmic_bytes_per_launch": alg_union_scaled, "achieved": alg_union_scaled / (dev_ms * 1e-3) / 1e9,
                                                  "frac": alg_union_scaled / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                  "note": "exact on %d of the %d queries (every 25th), scaled by SURVEY's figure of all queries" % (n_union, NQ)}
    if cpu is not None:
        out["cpu_baseline"] = cpu
    if embedded:
        return out
    print(json.dumps(out), flush=True)


def sparse_multi(args, K, S, device, rank, world, dist, rccl):
    """--mode all2all-sp --gpus N (BASELINE configs[3]): prefix-bucket shards, one per rank (weak scaling: genomes N x longer, rank r owns
    the buckets congruent to r mod N); every step = dense accumulation of the rank's partial matrix on its GPU, one RCCL reduce-scatter
    of the triangle in flat chunks (xGMI is point to point: every peer pair uses its own link), and the compaction of the rank's own
    chunk into sparse rows (kmdb_sparse_from_dense_device).  The ranks' rows concatenate to the reference's all2all_sp output
    (similarity_calculator.cpp:442-657, array.h:391-446); nothing but the sparse rows leaves the devices."""
    dev_index = device.index or 0
    N, cs, L, k, f = args.samples, args.clade_size, args.length * world, args.k, args.fraction
    arr, _, _, _, _ = generate_in_child(dev_index, n_samples=N, clade_size=cs, length=L, k=k, seed=args.seed, rank=rank, world=world, fraction=f)
    release_generator_memory(rank)
    d, upload_s = upload(K, arr, N, k, dev_index)
    cells = d.tri_size()
    per = (cells + world - 1) // world
    M = torch.zeros(per * world, dtype=torch.int32, device=device)          # the triangle, padded to equal chunks
    mine = torch.zeros(max(per, 1), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    lo, hi = min(cells, rank * per), min(cells, (rank + 1) * per)

    def step():
        d.all2all_dense_device(M.data_ptr(), stream=stream)
        dev_ms = d.stats()["kernel_ms"]
        if args.backend == "nccl":
            dist.reduce_scatter_tensor(mine, M, op=dist.ReduceOp.SUM)        # uint32 wrap-around sum == int32 sum bitwise
        else:
            torch.cuda.synchronize()
            h = M.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            mine.copy_(h[rank * per: (rank + 1) * per])
        sp = d.sparse_from_dense_device(mine.data_ptr(), lo, hi, stream=stream)
        return sp, dev_ms + d.stats()["kernel_ms"]

    for _ in range(args.warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    dev_ms = []
    for _ in range(args.steps):
        sp, ms = step()
        dev_ms.append(ms)
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    cdev = device if args.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([elapsed, upload_s, float(np.mean(dev_ms))], dtype=torch.float64, device=cdev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, upload_s, dev_mean = float(t[0]), float(t[1]), float(t[2])
    st = d.stats()
    tot = torch.tensor([float(st["sum_pairs"]), float(sp.nnz), float(sp.val.astype(np.uint64).sum()), float(pattern_bytes(arr))], dtype=torch.float64, device=cdev)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    sum_pairs, nnz, val_sum, bpat = (float(x) for x in tot)
    # identity over the whole output: the ranks' non-zeros sum to sum over all shards of sum_p w_p C(n_p, 2)
    assert val_sum == sum_pairs, "sharded sparse output checksum mismatch: %r vs %r" % (val_sum, sum_pairs)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        alg = bpat + 8 * nnz
        achieved = alg / (ms_per_step * 1e-3) / 1e9
        out = {"metric": "all2all-sp k-mer pair-comparisons/sec", "value": sum_pairs / (elapsed / args.steps), "unit": "kmer-pair-comparisons/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "u32", "data": "synthetic",
               "config": {"workload": "%s: %d synthetic %g Mbp genomes (clades of %d), k=%d f=%g, all2all-sp, k-mer space sharded by prefix bucket over %d GPUs + "
                                      "reduce-scatter of the partial matrices + per-rank compaction" % (args.workload, N, L / 1e6, cs, k, f, world),
                          "samples": N, "genome_length_bp": L, "k": k, "fraction": f, "mode": "all2all-sp", "nnz": int(nnz), "patterns_rank0": int(d.P),
                          "parallelism": "prefix-shard x%d" % world, "rccl": rccl, "collective": "reduce_scatter" if args.backend == "nccl" else "gloo all_reduce (functional test)"},
               "wall": {"upload_s": upload_s, "device_ms_per_step_max_rank": dev_mean,
                        "note": "ms_per_step: wall clock of one step on the slowest rank (dense accumulation + collective + compaction + D2H of the rank's sparse rows)"},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * world), "traffic": None,
                            "kernel": "whole step on every rank", "kernel_ms": ms_per_step, "algorithmic_bytes_per_launch": alg}}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def node_driver(args):
    """`--driver node`: the PRODUCT's multi-GPU path under the bench contract (reference call site made multi-GPU:
    src/console_all2all.cpp:31-36).  One process: the database (one, with its hashtable items — the shards are planned from them) is cut into
    --shards prefix-bucket shards by kmdb_node_upload, shard s on GPU s % N; a step = kmdb_node_all2all_dense: every device thread runs its
    shards, ONE ncclReduceScatter over flat chunks of the triangle (device-to-device over xGMI), every device brings its chunk to the host.
    Under a launcher (the driver starts N ranks) rank 0 does the work and the other ranks wait at the barriers: the N GPUs belong to the one
    process, as they do to `kmer-db-amd all2all -gpus N`.  The line has the schema of the other lines; per device: kmdb_node_device_stats."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")                      # (rendezvous of the launcher's ranks only: no tensor travels through it)
        if rank != 0:
            dist.barrier()
            dist.destroy_process_group()
            return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda is not available); there is no CPU path to time")
    n_dev = args.gpus
    if torch.cuda.device_count() < n_dev:
        raise SystemExit("bench.py: --gpus %d but only %d devices are visible" % (n_dev, torch.cuda.device_count()))
    n_shards = args.shards if args.shards else n_dev
    K = import_kmerdb_amd()
    torch.cuda.set_device(0)
    arr, names, counts, nk, items = generate_in_child(0, n_samples=args.samples, clade_size=args.clade_size, length=args.length, k=args.k, seed=args.seed,
                                                      rank=0, world=1, progress=100, with_items=n_shards > 1)
    release_generator_memory(0)
    view = K.make_view(args.k, args.samples, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"],
                       arr["last_sample_id"], arr["num_bits"], arr["data_offset"], arr["data"],
                       bucket_offset=None if items is None else items[0], slots=None if items is None else items[1])
    t0 = time.perf_counter()
    nd = K.NodeDB(view, n_shards, devices=list(range(n_dev)))
    upload_s = time.perf_counter() - t0
    st0 = nd.stats()
    log("node driver: %d shards on %d devices, upload %.2f s (host plan %.2f s), RCCL %s" % (st0["n_shards"], st0["n_devices"], upload_s, st0["plan_s"], st0["rccl_version"] or None))
    t0 = time.perf_counter()
    first = nd.all2all_dense()
    cold_ms = (time.perf_counter() - t0) * 1e3
    for _ in range(args.warmup):
        nd.all2all_dense()
    per_dev, M = [], first
    t0 = time.perf_counter()
    for _ in range(args.steps):
        M = nd.all2all_dense()
        per_dev.append(nd.stats()["devices"])
    elapsed = time.perf_counter() - t0
    # size-independent checks of the timed result: sum of the matrix == sum_p w_p C(n_p, 2) (the shards' partial matrices add up to the
    # whole database's), and warm == cold
    n_p = arr["num_samples"].astype(np.uint64)
    sum_pairs = int((arr["num_kmers"].astype(np.uint64) * (n_p * (n_p - np.uint64(1)) // np.uint64(2))).sum())
    assert int(M.astype(np.uint64).sum()) == sum_pairs, "matrix checksum mismatch: %d vs %d" % (int(M.astype(np.uint64).sum()), sum_pairs)
    assert np.array_equal(M, first), "warm call differs from the first call"
    cells = args.samples * (args.samples - 1) // 2
    alg = int((40 + 16 * ((arr["num_bits"].astype(np.int64) + 127) // 128)).sum()) + 4 * cells
    ms = elapsed / args.steps * 1e3
    stl = nd.stats()
    mean = lambda key: [float(np.mean([d[i][key] for d in per_dev])) for i in range(len(per_dev[0]))]      # noqa: E731
    v = stl["rccl_version"]
    rccl = None if not v else ("%d.%d.%d" % (v // 10000, v // 100 % 100, v % 100) if v >= 10000 else str(v))
    achieved = alg / (ms * 1e-3) / 1e9
    out = {
        "metric": "all2all k-mer pair-comparisons/sec", "value": sum_pairs / (elapsed / args.steps), "unit": "kmer-pair-comparisons/s",
        "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
        "config": {
            "workload": "%s: %d synthetic %g Mbp genomes (clade-mutation model, clades of %d, r1=0.10 r2=0.01), k=%d f=1.0, dense all2all, ONE database "
                        "in %d prefix-bucket shards over %d GPUs through the product's node driver (kmdb_node_upload / kmdb_node_all2all_dense: a host "
                        "thread per device, one ncclReduceScatter over flat chunks of the triangle, D2H of every device's chunk)"
                        % (args.workload, args.samples, args.length / 1e6, args.clade_size, args.k, n_shards, n_dev),
            "samples": args.samples, "genome_length_bp": args.length, "k": args.k, "fraction": 1.0, "driver": "node",
            "patterns_rank0": int(arr["num_kmers"].size), "distinct_kmers_rank0": nk, "parallelism": "prefix-shard x%d on %d devices" % (n_shards, n_dev),
            "rccl": rccl, "n_ranks_seen": stl["n_devices"],
            "per_rank": {"device": [d["device"] for d in stl["devices"]], "shards": [d["n_shards"] for d in stl["devices"]],
                         "call_ms": mean("call_ms"), "collective_ms": mean("collective_ms"), "d2h_ms": mean("d2h_ms"),
                         "patterns": [int(d["n_patterns"]) for d in stl["devices"]], "h2d_bytes": [int(d["h2d_bytes"]) for d in stl["devices"]],
                         "block_records": [int(d["n_records"]) for d in stl["devices"]], "upload_s": [d["upload_s"] for d in stl["devices"]],
                         "backend": "rccl (dlopen, ncclCommInitAll)" if rccl else "none (one device)",
                         "collective": "ncclReduceScatter(uint32, sum) + D2H of the device's chunk" if rccl else "none"},
            "sample_pairs_per_s": cells / (elapsed / args.steps), "path": "block-record pipeline",
        },
        "wall": {"upload_s": upload_s, "plan_s": stl["plan_s"], "cold_call_ms": cold_ms, "warm_ms": ms,
                 "note": "ms_per_step: host wall clock of kmdb_node_all2all_dense (slowest device's shards + the collective + the copy of its chunk to "
                         "the host matrix); upload_s: kmdb_node_upload (host shard plan + per device: narrowing, H2D, device layout, working set)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * n_dev, "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * n_dev), "traffic": None,
                     "kernel": "whole step: every device's all2all calls (one per shard) + reduce-scatter + D2H", "kernel_ms": ms, "algorithmic_bytes_per_launch": alg,
                     "per_kernel_ms": {"compute_slowest_device": float(max(mean("call_ms"))), "collective_slowest_device": float(max(mean("collective_ms"))),
                                       "d2h_slowest_device": float(max(mean("d2h_ms")))}},
    }
    nd.close()
    try:                                                     # (RCCL writes its banner through C stdio: out before the line, so that the line is the last one)
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def respawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run"""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--clade-size", type=int, default=None)
    ap.add_argument("--length", type=int, default=None, help="genome length (bp); per GPU with --scaling weak")
    ap.add_argument("--mode", default="all2all", choices=sorted(MODE_WORKLOAD),
                    help="all2all (default, the headline line) or one of the secondary rows: all2all-sp, new2all, db2db (1 GPU)")
    ap.add_argument("--k", type=int, default=None)
    ap.add_argument("--fraction", type=float, default=None)
    ap.add_argument("--r1", type=float, default=None, help="secondary modes: mutation rate between the root genome and a clade ancestor (0.75: independent roots)")
    ap.add_argument("--queries", type=int, default=None)
    ap.add_argument("--seed", type=int, default=20260928 + 1)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-sample-length", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sweep", default="short", choices=["short", "full"], help="reference baseline: -t 16 / 32 + -buffer 32 at the better one (short), or -t 16 / 32 / 64 / 128 (full)")
    ap.add_argument("--no-extra", action="store_true", help="default workload only: skip the 10 000-sample workload that rides along in the same JSON line")
    ap.add_argument("--generate-spec", default=None, help=argparse.SUPPRESS)       # generate_in_child()'s other side
    ap.add_argument("--tmp", default=None, help="directory for the reference's .db files (default: the system temp dir)")
    ap.add_argument("--collective", default="reduce", choices=["reduce", "reduce_scatter"],
                    help="how the partial matrices of --gpus N meet: one reduce to rank 0 (north_star), or a reduce-scatter in flat chunks of the "
                         "triangle after which every rank brings its own chunk to the host (SURVEY 8e: direct, every peer pair over its own xGMI link)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: functional test of the multi-rank path on a box with fewer GPUs than ranks "
                         "(ranks share devices, the matrix reduce goes through host memory); never used for reported numbers")
    ap.add_argument("--driver", default="auto", choices=["auto", "node", "ranks"],
                    help="node: ONE process drives the devices through the product's own multi-GPU driver (kmdb_node_upload / kmdb_node_all2all_dense, "
                         "csrc/node.hip: prefix-bucket shards, one host thread per device, ONE ncclReduceScatter) — what `kmer-db-amd all2all -gpus N` "
                         "runs; ranks: one process per GPU under torch.distributed (kmdb_all2all_dense_device + an RCCL collective issued from Python). "
                         "auto = node for --gpus > 1 in the all2all mode, ranks otherwise")
    ap.add_argument("--shards", type=int, default=None, help="--driver node: prefix shards of the database (default: one per GPU; more than GPUs: several per "
                                                             "device, one after the other — `--gpus 1 --shards 8` is the one-GPU anchor of an 8-GPU run)")
    ap.add_argument("--one-rank-group", action="store_true",
                    help="--gpus 1 only: run the multi-rank code path (process group, collective of every step, per-rank figures) on a ONE-rank "
                         "RCCL group — what a one-GPU box can exercise of the --gpus N path; a functional check, never a reported number")
    args = ap.parse_args()
    if args.generate_spec is not None:
        generator_child(json.loads(args.generate_spec))
        return
    if args.workload is None:
        # one GPU: BASELINE configs[1]; several: configs[2] — 10 000 samples, every rank one GPU's share of the 5 Mbp genomes (c3gpu =
        # 5 Mbp / 8: at --gpus 8 with the default --scaling weak the job IS configs[2])
        args.workload = "c3gpu" if (args.mode == "all2all" and args.gpus > 1) else MODE_WORKLOAD[args.mode]
    for key, val in dict(dict(k=18, fraction=1.0, queries=0, r1=0.10), **WORKLOADS[args.workload]).items():
        if getattr(args, key, None) is None:
            setattr(args, key, val)

    if args.driver == "auto":
        # (--backend gloo is the functional test of the rank form on a box with fewer GPUs than ranks: it keeps the rank form)
        args.driver = "node" if (args.gpus > 1 and args.mode == "all2all" and not args.one_rank_group and args.backend == "nccl") else "ranks"
    if args.driver == "node":
        if args.mode != "all2all":
            raise SystemExit("bench.py: --driver node is for the all2all mode")
        node_driver(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_ranks(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    if args.one_rank_group and world != 1:
        raise SystemExit("bench.py: --one-rank-group is for --gpus 1")
    multi = world > 1 or args.one_rank_group              # the process group and the collectives are used
    if args.one_rank_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
        os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda is not available); there is no CPU path to time")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    rccl = None
    if multi:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)      # RCCL over xGMI
            try:
                rccl = ".".join(str(x) for x in torch.cuda.nccl.version())
            except Exception:                                      # reporting only
                rccl = "unknown"
        else:
            dist.init_process_group("gloo")
    K = import_kmerdb_amd()
    import importlib
    S = importlib.import_module("kmerdb_amd.synth")

    if args.mode != "all2all":
        if multi and args.mode == "all2all-sp":
            sparse_multi(args, K, S, device, rank, world, dist, rccl)
            return
        if world != 1:
            raise SystemExit("bench.py: --mode %s runs on one GPU" % args.mode)
        secondary_mode(args, K, S, device)
        return
    if args.fraction != 1.0:
        raise SystemExit("bench.py: the all2all mode runs at f = 1.0 (use --mode all2all-sp for minhash databases)")
    strong = world > 1 and args.scaling == "strong"
    total_len = args.length if (world == 1 or strong) else args.length * world
    if strong:
        # every rank derives the same database and keeps its prefix shard of it
        arr, names, counts, nk, items = generate_in_child(dev_index, n_samples=args.samples, clade_size=args.clade_size, length=total_len, k=args.k, seed=args.seed,
                                                          rank=rank, world=1, progress=100 if rank == 0 else None, with_items=True)
        release_generator_memory(rank)
        db, upload_s = upload(K, arr, args.samples, args.k, dev_index, items=items, prefix_shard=(rank, world))
    else:
        arr, names, counts, nk, items = generate_in_child(dev_index, n_samples=args.samples, clade_size=args.clade_size, length=total_len, k=args.k, seed=args.seed,
                                                          rank=rank, world=world, progress=100 if rank == 0 else None)
        release_generator_memory(rank)
        db, upload_s = upload(K, arr, args.samples, args.k, dev_index)
    st0 = db.stats()
    log("[rank %d] upload %.2f s; block width %d" % (rank, upload_s, st0["width"]))
    cells = db.tri_size()
    # the first call: host matrix out, as the front-end uses it (H2D / D2H inclusive, grid sizes measured on the way)
    t0 = time.perf_counter()
    first = db.all2all_dense()
    cold_ms = (time.perf_counter() - t0) * 1e3
    stc = db.stats()
    log("[rank %d] cold call %.1f ms (device pipeline %.2f ms, path %d)" % (rank, cold_ms, stc["kernel_ms"], stc["path"]))

    scatter = multi and args.collective == "reduce_scatter"
    per = (cells + world - 1) // world if scatter else 0
    M = torch.zeros(max(per * world if scatter else cells, 1), dtype=torch.int32, device=device)      # (reduce_scatter: the triangle padded to equal chunks)
    mine = torch.zeros(max(per, 1), dtype=torch.int32, device=device) if scatter else None
    hband = torch.zeros(max(per, 1), dtype=torch.int32).pin_memory() if scatter else None
    stream = torch.cuda.current_stream().cuda_stream

    coll_ev = []                                             # (before, after) events around the collective of every timed step

    def step():
        db.all2all_dense_device(M.data_ptr(), stream=stream)
        if multi:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            if scatter:
                # direct reduce-scatter in flat chunks of the triangle: xGMI is point to point, every peer pair sums its chunk over its own
                # link; every rank then brings ITS chunk to the host (the front-end writes the rows of its chunk)
                if args.backend == "nccl":
                    dist.reduce_scatter_tensor(mine, M, op=dist.ReduceOp.SUM)
                else:
                    torch.cuda.synchronize()
                    h = M.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    mine.copy_(h[rank * per: (rank + 1) * per])
                hband.copy_(mine)
            elif args.backend == "nccl":
                dist.reduce(M, dst=0, op=dist.ReduceOp.SUM)      # uint32 wrap-around sum == int32 sum bitwise
            else:
                torch.cuda.synchronize()
                h = M.cpu()
                dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    M.copy_(h)
            ev[1].record()
            coll_ev.append(ev)

    def fence():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    coll_ev.clear()
    call_ms, parts = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        _s = db.stats()
        call_ms.append(_s["kernel_ms"])
        parts.append((_s["k0_ms"], _s["k1n_ms"], _s["k1g_ms"], _s["k2_ms"]))
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        cdev = device if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed, upload_s, cold_ms], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, upload_s, cold_ms = float(t[0]), float(t[1]), float(t[2])
        tot = torch.tensor([st0["sum_pairs"], st0["tree_updates"], st0["algorithmic_bytes"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        sum_pairs, tree_updates = float(tot[0]), float(tot[1])
        # every rank's own figures: device time of its calls (HIP events around the whole call) and of the collective (events around the
        # RCCL op on the stream it runs on; with --collective reduce_scatter the D2H of the rank's chunk is inside)
        mine_t = torch.tensor([float(np.mean(call_ms)), float(np.mean([a.elapsed_time(b) for a, b in coll_ev])), float(db.P), 1.0], dtype=torch.float64, device=cdev)
        per_rank = [torch.zeros_like(mine_t) for _ in range(world)]
        dist.all_gather(per_rank, mine_t)
        per_rank = [[float(x) for x in t.tolist()] for t in per_rank]
    else:
        sum_pairs, tree_updates = float(st0["sum_pairs"]), float(st0["tree_updates"])
        per_rank = None

    # size-independent check of the timed result: sum of the matrix == sum_p w_p C(n_p,2); and warm == cold
    if scatter:
        lo_c, hi_c = min(cells, rank * per), min(cells, (rank + 1) * per)
        part_sum = torch.tensor([float(hband[: hi_c - lo_c].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item())], dtype=torch.float64, device=cdev)
        dist.all_reduce(part_sum, op=dist.ReduceOp.SUM)
        assert float(part_sum[0]) == sum_pairs, "matrix checksum mismatch over the ranks' chunks: %r vs %r" % (float(part_sum[0]), sum_pairs)
    elif rank == 0:
        got = int(M[:cells].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) if cells else 0
        assert got == int(sum_pairs), "matrix checksum mismatch: %d vs %d" % (got, int(sum_pairs))
        if world == 1:
            assert np.array_equal(M[:cells].cpu().numpy().view(np.uint32), first), "warm call differs from the first call"

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        stl = db.stats()
        path_name = {1: "block-record pipeline", 2: "v1 tile kernel", 3: "v1 HBM-atomics kernel"}.get(stl["path"], "none")
        kern_ms = float(np.mean(call_ms))                  # HIP events around the WHOLE call on its stream: zeroing, decode, emit, apply
        pk = np.mean(np.array(parts), axis=0)
        alg = st0["algorithmic_bytes"]
        achieved = alg / (kern_ms * 1e-3) / 1e9
        # HBM bytes per call come from separate rocprofv3 --pmc runs of this same command (profiles/): they
        # cannot be collected from inside the timed process; quoted only for the default workload
        traffic, traffic_src = None, None
        if not multi and args.length == WORKLOADS[args.workload]["length"] and args.samples == WORKLOADS[args.workload]["samples"]:
            traffic, traffic_src = replayed_traffic(args.workload)
        out = {
            "metric": "all2all k-mer pair-comparisons/sec",
            "value": sum_pairs / (elapsed / args.steps),
            "unit": "kmer-pair-comparisons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d synthetic %g Mbp genomes (clade-mutation model, clades of %d, r1=0.10 r2=0.01), k=%d f=1.0, "
                            "dense all2all%s" % (args.workload, args.samples, total_len / 1e6, args.clade_size, args.k,
                                                 "" if not multi else ", k-mer space sharded by prefix bucket over %d GPUs (%s) + RCCL %s"
                                                 % (world, "one database, kmdb_db_upload_shard" if strong else "per-rank databases",
                                                    "reduce-scatter by flat chunks + D2H of every rank's chunk" if scatter else "reduce")),
                "samples": args.samples, "genome_length_bp": total_len, "k": args.k, "fraction": 1.0,
                "patterns_rank0": db.P, "distinct_kmers_rank0": nk, "parallelism": "prefix-shard x%d" % world, "rccl": rccl,
                "n_ranks_seen": world if per_rank is None else int(sum(r[3] for r in per_rank)),
                "per_rank": None if per_rank is None else {"call_ms": [r[0] for r in per_rank], "collective_ms": [r[1] for r in per_rank],
                                                           "patterns": [int(r[2]) for r in per_rank], "backend": args.backend,
                                                           "collective": "reduce_scatter + D2H of the rank's chunk" if scatter else "reduce to rank 0"},
                "sample_pairs_per_s": args.samples * (args.samples - 1) / 2 / (elapsed / args.steps),
                "cell_updates_per_s": tree_updates / (elapsed / args.steps),
                "path": path_name, "block_width": stl["width"],
            },
            "wall": {
                "upload_s": upload_s, "cold_call_ms": cold_ms, "cold_total_s": upload_s + cold_ms * 1e-3, "warm_ms": ms_per_step,
                "note": "upload = kmdb_db_upload (format conversion: host narrowing + H2D + device DFS layout; no sample id decoded except "
                        "the 1-in-%d sample of the block-width estimate); cold call = first kmdb_all2all_dense incl. D2H of the matrix; every "
                        "call, warm or cold, decodes, places and accumulates everything itself; the upload's host staging buffers (3.5 GB at c2) are "
                        "given back by a helper thread after the first call, not inside upload (their pages dropped on several threads under the "
                        "shared address-space lock, then unmapped); frontend_*: `kmer-db-amd all2all full.db out.csv` as a process on the "
                        "reference's own .db file — load / upload / compute / csv as it prints them, process_up = kernel start of the process to "
                        "the table on disk (incl. waiting for its page drops), frontend_s - process_up = the end of the process" % max(1, min(1024, db.P // 65536)),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "whole call: k0_decode_kernel x2 + k1n_kernel + wide list + k1w_kernel + cs_hist / cs_scatter (counting sort) + k2_jobs_kernel, with k2d_kernel (the slices' first-block records) on a side stream (+ zeroing, pool init)",
                "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg,
                "per_kernel_ms": {"decode": float(pk[0]), "emit_narrow": float(pk[1]), "wide_list+emit_wide": float(pk[2]), "apply": float(pk[3]),
                                  "whole_call": kern_ms},
                "cold_call_frac": alg / (cold_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "measured_copy_GBs": measured_copy_gbs(device),      # what a plain copy reaches on this box (peak above is the nominal figure)
                "block_records_per_launch": stl["n_records"], "first_block_records_per_launch": stl["n_direct"], "wide_nodes": stl["n_wide"], "nodes_joined_per_tile": stl["n_joined"],
                "record_chunks": stl["n_chunks"],
            },
        }
        if not multi and not args.no_cpu_baseline:
            from oracle import oracle as O
            db.close()
            out["cpu_baseline"] = cpu_baseline(K, S, O, args, device, arr, names, counts, nk, first)
            cb = out["cpu_baseline"]
            if cb["kind"] == "reference":
                out["wall"]["reference_compute_s"] = cb["seconds"]
                # end to end, both as whole processes on the same .db file: the reference's compute-only run (it writes no CSV here) and the
                # front-end's read + upload + call + CSV
                out["wall"]["reference_process_s"] = cb["process_seconds"]
                out["wall"].update(cb.pop("frontend"))
        if not multi and args.workload == "c2" and args.length == WORKLOADS["c2"]["length"] and args.samples == WORKLOADS["c2"]["samples"] and not args.no_extra:
            # the 10 000-sample workload rides along in the same line (the configuration BASELINE's target is written for)
            db.close()
            del arr
            torch.cuda.empty_cache()
            out["extra"] = {"c3part": extra_workload(K, S, args, device, "c3part")}
            torch.cuda.empty_cache()
            # one GPU's share of BASELINE configs[2] itself (10 000 x 625 kbp), under the same clock
            out["extra"]["c3gpu"] = extra_workload(K, S, args, device, "c3gpu", reference=False, definition_rows=8)
            # the secondary rows ride along too, so that the driver's clock times them: new2all at configs[4]'s shape
            # (1000 queries against 10 000 samples) and one all2all-parts cell, every row compared with the real reference where it is built
            import copy
            for key, mode, wl in (("new2all_c5part", "new2all", "c5part"), ("db2db_parts", "db2db", "parts")):
                torch.cuda.empty_cache()
                a2 = copy.copy(args)
                a2.mode, a2.workload = mode, wl
                for kk, vv in dict(dict(k=18, fraction=1.0, queries=0, r1=0.10), **WORKLOADS[wl]).items():
                    setattr(a2, kk, vv)
                a2.steps, a2.warmup = 5, 2
                t_sec = time.time()
                res = secondary_mode(a2, K, S, device, embedded=True)
                res["seconds_in_bench"] = time.time() - t_sec
                out["extra"][key] = res
                log("%s: %.2f ms per call, frac %.4f, reference %s (%.0f s in the bench)" % (key, res["ms_per_step"], res["roofline"]["frac"],
                                                                                           res.get("cpu_baseline", {}).get("kind"), res["seconds_in_bench"]))
            log("c3gpu: %.2f ms per call, frac %.4f, %d block records, %d nodes joined per tile, rows_from_definition: %s" % (
                out["extra"]["c3gpu"]["ms_per_step"], out["extra"]["c3gpu"]["roofline"]["frac"], out["extra"]["c3gpu"]["records"],
                out["extra"]["c3gpu"]["nodes_joined_per_tile"], out["extra"]["c3gpu"]["rows_from_definition"]))
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
