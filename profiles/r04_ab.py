#!/usr/bin/env python3
"""Same-box, same-database A/B of run-time knobs of the all2all call (environment variables the engine reads per call): the synthetic
database is generated and uploaded ONCE, then every variant runs warm calls on the same handle.  Prints whole-call ms (HIP events
around the call), the stages, and whether the matrix equals the first variant's (timing experiments that switch work off differ
on purpose).

    python profiles/r04_ab.py c3part "" "KMDB_K1W_DEBUG=1" "KMDB_K1W_DEBUG=2" ...
    python profiles/r04_ab.py c2 --reupload "" "KMDB_BLOCK_WIDTH=64"        (knobs read at upload: a fresh handle per variant)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    reupload = "--reupload" in sys.argv
    steps = 5
    for a in sys.argv[1:]:
        if a.startswith("--steps="):
            steps = int(a.split("=")[1])
    wl = B.WORKLOADS[args[0]]
    variants = args[1:] or [""]
    K = B.import_kmerdb_amd()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    length = wl["length"]
    for a in sys.argv[1:]:
        if a.startswith("--length="):
            length = int(a.split("=")[1])
    # the generated arrays are kept in shared memory for the other processes of the same job (compile-time variants: one process each)
    cache = "/dev/shm/kmdb_r04_ab_%s_%d" % (args[0], length)
    if os.path.isdir(cache) and os.path.exists(os.path.join(cache, "done")):
        arr = {nm[:-4]: np.load(os.path.join(cache, nm)) for nm in os.listdir(cache) if nm.endswith(".npy")}
    else:
        arr, names, counts, nk, _ = B.generate_in_child(0, n_samples=wl["samples"], clade_size=wl["clade_size"], length=length, k=18, seed=20260929, rank=0, world=1)
        try:
            os.makedirs(cache, exist_ok=True)
            for nm, a in arr.items():
                np.save(os.path.join(cache, nm + ".npy"), a)
            open(os.path.join(cache, "done"), "w").close()
        except OSError:
            pass
    db = None
    ref = None
    out = []
    for v in variants:
        saved = {}
        for kv in v.split():
            k, _, val = kv.partition("=")
            saved[k] = os.environ.get(k)
            os.environ[k] = val
        if db is None or reupload:
            if db is not None:
                db.close()
            db, up = B.upload(K, arr, wl["samples"], 18, 0)
        cells = db.tri_size()
        M = torch.zeros(max(cells, 1), dtype=torch.int32, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            db.all2all_dense_device(M.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        ms, parts = [], []
        for _ in range(steps):
            db.all2all_dense_device(M.data_ptr(), stream=stream)
            s = db.stats()
            ms.append(s["kernel_ms"])
            parts.append((s["k0_ms"], s["k1n_ms"], s["k1g_ms"], s["k2_ms"]))
        torch.cuda.synchronize()
        got = M[:cells].cpu().numpy()
        if ref is None:
            ref = got
        st = db.stats()
        pk = np.mean(np.array(parts), axis=0)
        line = {"variant": v or "default", "ms": round(float(np.mean(ms)), 3), "min_ms": round(float(np.min(ms)), 3), "decode": round(float(pk[0]), 3), "narrow": round(float(pk[1]), 3),
                "wide": round(float(pk[2]), 3), "sort_apply": round(float(pk[3]), 3), "records": st["n_records"], "width": st["width"],
                "equal_to_first": bool(np.array_equal(got, ref)), "checksum_ok": int(got.view(np.uint32).astype(np.uint64).sum()) == int(st["sum_pairs"])}
        print(json.dumps(line), flush=True)
        out.append(line)
        del M
        for k, val in saved.items():
            if val is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = val
    if db is not None:
        db.close()


if __name__ == "__main__":
    main()
