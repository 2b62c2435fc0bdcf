#!/usr/bin/env python3
"""bench.py — all2all common-k-mer counting throughput on MI355X.

One "step" = one dense all2all pass (SimilarityCalculator::all2all's job, reference
src/similarity_calculator.cpp:42-438) over a synthetic clade-mutation database that is already
resident in HBM: the block-record pipeline (gamma decode, narrow / wide emit, apply) + (N>1) the
RCCL sum of the per-GPU partial matrices.  Workload at N=1 is BASELINE.json configs[1]: 1000 synthetic 5 Mbp genomes,
k=18, f=1.0.  With --gpus N the k-mer space is sharded by prefix bucket (kmer >> 32, reference
src/types.h:25-27): the genomes are N x 5 Mbp long and rank r owns the k-mers whose bucket is
congruent to r mod N, so per-GPU work stays fixed ("weak") and the partial matrices sum exactly.

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


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_shard_db(K, S, n_samples, clade_size, length, k, seed, device, rank, world, progress=None):
    """patterns of the k-mers whose prefix bucket is owned by `rank`"""
    g = S.CladeGenomes(n_samples, clade_size, length, seed=seed, device=device)

    def kmers(i):
        km = S.kmers_of(g.sample(i), k)
        if world > 1:
            km = km[((km >> 32) % world) == rank]
        return km
    t0 = time.time()
    pat = S.build_patterns(kmers, n_samples, device, progress=progress)
    arr = S.to_view_arrays(pat)
    log("[rank %d] synth db: %d samples x %d bp, %d k-mers, %d patterns in %.1f s" % (
        rank, n_samples, length, pat["dictionary"].numel(), arr["num_kmers"].size, time.time() - t0))
    names = [g.name(i) for i in range(n_samples)]
    return arr, names, pat["sample_counts"], int(pat["dictionary"].numel())


def upload(K, arr, n_samples, k, device_index):
    t0 = time.time()
    view = K.make_view(k, n_samples, arr["num_kmers"], arr["parent_id"], arr["num_samples"], arr["num_local"],
                       arr["last_sample_id"], arr["num_bits"], arr["data_offset"], arr["data"])
    d = K.DeviceDB(view, device=device_index)
    log("  layout + upload to HBM: %.1f s" % (time.time() - t0))
    return d


def cpu_baseline(K, S, args, device):
    """The real reference hot path (oracle/_ref) — or the oracle's C restatement — timed on the
    host cores on a bounded sample: the same 1000-sample model at a shorter genome length."""
    from oracle import oracle as O
    L = args.cpu_sample_length
    arr, names, counts, nk = build_shard_db(K, S, args.samples, args.clade_size, L, args.k, args.seed, device, 0, 1)
    cores = os.cpu_count() or 1
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sample.db")
        S.write_db(path, args.k, 1.0, names, counts, arr, kmers_count=nk)
        odb = O.OracleDB(path, skip_hashtables=True)
        uc = odb.update_counts()
        d = upload(K, arr, args.samples, args.k, device.index or 0)
        gpu = d.all2all_dense()
        if O.have_ref():
            # the reference's 4-stage pipeline does not scale to hundreds of threads on inputs of this
            # size; sweep -t / -buffer (README.md:185 of the reference) and report its best run
            best, tried = None, []
            for thr in sorted({min(cores, t) for t in (8, 16, 32, 64, 128)}):
                for buf in (8, 32):
                    m, info = O.ref_all2all(path, os.path.join(td, "m.u32"), threads=thr, buffer_mb=buf)
                    tried.append((thr, buf, round(info["seconds"], 3)))
                    if best is None or info["seconds"] < best[1]["seconds"]:
                        best = (m, info)
                    if info["seconds"] > 20:
                        break
            m, info = best
            log("  reference sweep (threads, bufferMb, s):", tried)
            kind, secs, used = "reference", info["seconds"], info["threads"]
        else:
            t0 = time.time()
            m = odb.all2all_dense()
            kind, secs, used = "port", time.time() - t0, 1
        assert np.array_equal(m, gpu), "GPU result differs from the CPU baseline on the sample database"
        d.close()
    return {
        "value": uc["sum_matrix"] / secs, "unit": "kmer-pair-comparisons/s", "cores": used, "host_cores": cores, "kind": kind,
        "seconds": secs, "cell_updates_per_s": uc["tree_updates"] / secs,
        "sample": "same %d-sample clade model at genome length %d bp (1/%d of the timed workload), %d patterns; "
                  "GPU matrix on this sample verified bit-identical" % (args.samples, L, max(1, args.length // L), odb.P),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--clade-size", type=int, default=50)
    ap.add_argument("--length", type=int, default=5_000_000, help="genome length per GPU (bp)")
    ap.add_argument("--k", type=int, default=18)
    ap.add_argument("--seed", type=int, default=20260928 + 1)
    ap.add_argument("--cpu-sample-length", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: functional test of the multi-rank path on a box with fewer GPUs than ranks "
                         "(ranks share devices, the matrix reduce goes through host memory); never used for reported numbers")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda is not available); there is no CPU path to time")
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)      # RCCL over xGMI
        else:
            dist.init_process_group("gloo")
    K = import_kmerdb_amd()
    import importlib
    S = importlib.import_module("kmerdb_amd.synth")

    total_len = args.length * world
    arr, names, counts, nk = build_shard_db(K, S, args.samples, args.clade_size, total_len, args.k, args.seed, device,
                                            rank, world, progress=100 if rank == 0 else None)
    torch.cuda.empty_cache()
    db = upload(K, arr, args.samples, args.k, dev_index)
    del arr
    st0 = db.stats()
    cells = db.tri_size()
    M = torch.zeros(max(cells, 1), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        db.all2all_dense_device(M.data_ptr(), stream=stream)
        if world > 1:
            if args.backend == "nccl":
                dist.reduce(M, dst=0, op=dist.ReduceOp.SUM)      # uint32 wrap-around sum == int32 sum bitwise
            else:
                torch.cuda.synchronize()
                h = M.cpu()
                dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)
                if rank == 0:
                    M.copy_(h)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    dom_ms, pipe_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        _s = db.stats()
        dom_ms.append(_s["dominant_kernel_ms"])
        pipe_ms.append(_s["k0_ms"] + _s["k1_ms"] + _s["k2_ms"])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        cdev = device if args.backend == "nccl" else torch.device("cpu")
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([st0["sum_pairs"], st0["tree_updates"], st0["algorithmic_bytes"]], dtype=torch.float64, device=cdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        sum_pairs, tree_updates = float(tot[0]), float(tot[1])
    else:
        sum_pairs, tree_updates = float(st0["sum_pairs"]), float(st0["tree_updates"])

    # size-independent check of the timed result: sum of the matrix == sum_p w_p C(n_p,2)
    if rank == 0:
        got = int(M[:cells].to(torch.int64).bitwise_and(0xFFFFFFFF).sum().item()) if cells else 0
        assert got == int(sum_pairs), "matrix checksum mismatch: %d vs %d" % (got, int(sum_pairs))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        stl = db.stats()
        if stl["n_records"]:
            # block-record pipeline: the kernels share the pass; the roofline is quoted on their SUM
            # (decode + narrow/wide emit + apply), never on the longest one alone
            names_ms = [("b3_decode_kernel" if stl["k0_ms"] > 0 else None, stl["k0_ms"]),
                        ("b3_narrow_kernel+b3_emit_kernel" if stl["k0_ms"] > 0 else "b2_emit_kernel", stl["k1_ms"]),
                        ("b2_apply_kernel", stl["k2_ms"])]
            kern_ms = float(np.mean(pipe_ms))
            dom_name = "+".join(n for n, _ in names_ms if n)
        else:
            kern_ms = float(np.mean(dom_ms))
            dom_name = "a2a_tile_kernel"
        alg = st0["algorithmic_bytes"]
        achieved = alg / (kern_ms * 1e-3) / 1e9
        # HBM bytes per pass come from separate rocprofv3 --pmc runs of this same command (profiles/): they
        # cannot be collected from inside the timed process; quoted only for the default workload
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
        if os.path.exists(tpath) and world == 1 and args.length == 5_000_000 and args.samples == 1000:
            with open(tpath) as f:
                tj = json.load(f)
            traffic, traffic_src = tj["traffic_bytes_per_pass"], "profiles/latest_traffic.json: " + tj["source"]
        out = {
            "metric": "all2all k-mer pair-comparisons/sec",
            "value": sum_pairs / (elapsed / args.steps),
            "unit": "kmer-pair-comparisons/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": "%d synthetic %g Mbp genomes (clade-mutation model, clades of %d, r1=0.10 r2=0.01), k=%d f=1.0, "
                            "dense all2all%s" % (args.samples, total_len / 1e6, args.clade_size, args.k,
                                                 "" if world == 1 else ", k-mer space sharded by prefix bucket over %d GPUs + RCCL reduce" % world),
                "samples": args.samples, "genome_length_bp": total_len, "k": args.k, "fraction": 1.0,
                "patterns_rank0": db.P, "distinct_kmers_rank0": nk, "parallelism": "prefix-shard x%d" % world,
                "sample_pairs_per_s": args.samples * (args.samples - 1) / 2 / (elapsed / args.steps),
                "cell_updates_per_s": tree_updates / (elapsed / args.steps),
                "tile_flushes": db.stats()["tile_flushes"],
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": dom_name, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg,
                "per_kernel_ms": {"decode": stl["k0_ms"], "emit": stl["k1_ms"], "apply": stl["k2_ms"], "whole_call": stl["kernel_ms"]},
                "block_records_per_launch": stl["n_records"],
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(K, S, args, device)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
