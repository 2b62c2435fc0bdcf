#!/usr/bin/env python3
"""Reduce rocprofv3 --pmc per-dispatch CSVs (profiles/collect_counters.sh) to one row per kernel of the timed calls:
<out>/<tag>_counters.json (raw means per launch) and <out>/<tag>_counters.md (derived: active-lane %, VALU utilisation,
stall shares, occupancy).  Only the dispatches after the last upload kernel count (bench.py --steps 2 --warmup 1 makes
four calls: cold, one warm-up, two timed), and a kernel's row is the mean over its launches.

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" and the s_memtime row): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are in
quad-cycles summed over the waves; SQ_BUSY_CYCLES is per shader engine; FETCH_SIZE / WRITE_SIZE are KB (FETCH_SIZE reports
half of a wide coalesced read on gfx950)."""
import csv, json, os, sys
from collections import defaultdict

tag, out = sys.argv[1:3]
files = sys.argv[3:]          # per-dispatch CSVs; none: the table is re-made from <out>/<tag>_counters.json
KERNELS = ("k0_decode_kernel", "k0_short_direct_kernel", "l2_ranks_kernel", "l2_lists_kernel", "l2_join_apply_kernel", "k2j_build_kernel", "k2_jobs_kernel", "k2d_kernel", "k1n_kernel", "k1g_kernel", "k1w_kernel", "k2_apply_kernel", "k2_sorted_kernel", "cs_hist_kernel", "cs_scatter_kernel",
           "rs_hist_kernel", "rs_scatter_kernel", "rs_rows_kernel", "rg_hist_kernel", "rg_scatter_kernel", "ct_hist_kernel", "ct_scatter_kernel", "wrun_anc_kernel",
           "wide_count_kernel", "wide_expand_kernel", "n2a_walk_kernel", "n2a_probe_kernel", "n2a_extract_kernel", "d2_emit_kernel", "d2_probe_kernel", "row_nnz_kernel",
           "row_compact_kernel")


def short(name):
    for k in KERNELS:
        if k in name:
            if k == "k0_decode_kernel":
                return "k0_decode_kernel<long>" if "<true>" in name or "true" in name.split("k0_decode_kernel")[1][:12] else "k0_decode_kernel<short>"
            return k
    if "rocprim" in name and ("radix" in name or "onesweep" in name):
        return "rocprim radix sort"
    if "rocprim" in name and "scan" in name:
        return "rocprim scan"
    return None


vals = defaultdict(lambda: defaultdict(float))      # kernel -> counter -> sum
cnts = defaultdict(lambda: defaultdict(int))
meta = {}                                            # kernel -> (vgpr, lds, scratch, wg)
for path in files:
    rows = list(csv.DictReader(open(path)))
    if not rows:
        continue
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    last_upload = -1
    for i, r in enumerate(rows):
        n = r.get("Kernel_Name", "")
        if "lay_" in n or "width_estimate" in n or "iota_u32" in n or "pair_estimate" in n:
            last_upload = i
    for r in rows[last_upload + 1:]:
        k = short(r.get("Kernel_Name", ""))
        if k is None:
            continue
        c = r["Counter_Name"]
        vals[k][c] += float(r["Counter_Value"])
        cnts[k][c] += 1
        meta[k] = {"vgpr": r.get("VGPR_Count") or r.get("Arch_VGPR_Count"), "accum_vgpr": r.get("Accum_VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                   "lds_bytes": r.get("LDS_Block_Size"), "scratch_bytes": r.get("Scratch_Size"), "workgroup": r.get("Workgroup_Size"), "grid": r.get("Grid_Size")}

res = {}
for k in vals:
    res[k] = {c: vals[k][c] / max(1, cnts[k][c]) for c in vals[k]}
    res[k]["_launches_counted"] = max(cnts[k].values())
    res[k]["_meta"] = meta.get(k, {})
if files:
    json.dump(res, open(os.path.join(out, tag + "_counters.json"), "w"), indent=1)
else:
    res = json.load(open(os.path.join(out, tag + "_counters.json")))
# HBM traffic of one call (every kernel of the call that has both counters), stamped with the hash of the sources it was measured on:
# bench.py replays it into its line only when the code is unchanged
if files and any("FETCH_SIZE" in d and "WRITE_SIZE" in d for d in res.values()):
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmer-db_amd", "csrc")
    for fn in ("a2a_blocks.hip", "a2a_v1.hip", "device_common.h", "engine.hip", "engine_internal.h", "engine_state.h", "layout.hip", "prim.h"):   # = bench.py ALL2ALL_SOURCES
        h.update(fn.encode())
        h.update(open(os.path.join(csrc, fn), "rb").read())
    # per call: a kernel's mean per launch x its launches per call (launches counted / 4 calls: cold, warm-up, two timed)
    fb = {k: d["FETCH_SIZE"] * 1024.0 * cnts[k]["FETCH_SIZE"] / 4.0 for k, d in res.items() if "FETCH_SIZE" in d}
    wb = {k: d["WRITE_SIZE"] * 1024.0 * cnts[k]["WRITE_SIZE"] / 4.0 for k, d in res.items() if "WRITE_SIZE" in d}
    tj = {"sources_sha16": h.hexdigest()[:16],
          "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (--kernel-trace) of python bench.py --no-cpu-baseline --steps 2 --warmup 1 "
                    "(profiles/collect_counters.sh), mean over its 4 all2all calls, the engine's own kernels + rocprim sorts / scans (memsets not "
                    "included); counters are KB, x1024; FETCH_SIZE can under-count wide coalesced reads by up to 2x on gfx950 "
                    "(MI355X_MICROARCH.md, HBM section), so this is a lower bound",
          "fetch_bytes": fb, "write_bytes": wb, "traffic_bytes_per_pass": sum(fb.values()) + sum(wb.values())}
    json.dump(tj, open(os.path.join(out, tag + "_traffic.json"), "w"), indent=1)


def g(d, c):
    return d.get(c, float("nan"))


lines = ["| kernel | waves | VGPR / LDS B / scratch | active lanes % | VALU busy % of wave time | waiting (s_waitcnt / barrier) % | issue stall % | LDS stall % of issue stall | "
         "LDS conflict % | VALU : SALU : LDS : VMEM : SMEM insts per wave | MFMA busy % | fetch MB | write MB |", "|" + "---|" * 14]
for k in sorted(res):
    d = res[k]
    waves = g(d, "SQ_WAVES")
    wc = g(d, "SQ_WAVE_CYCLES")
    lanes = 100.0 * g(d, "SQ_THREAD_CYCLES_VALU") / (64.0 * g(d, "SQ_ACTIVE_INST_VALU")) if d.get("SQ_ACTIVE_INST_VALU") else float("nan")
    # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = active lanes per VALU instruction (a full-wave copy kernel measures 62 of 64);
    # SQ_ACTIVE_INST_VALU counts one quad-cycle per wave64 VALU instruction (it equals SQ_INSTS_VALU within 1 %)
    valu_busy = 100.0 * g(d, "SQ_ACTIVE_INST_VALU") / wc if wc == wc and wc else float("nan")
    wait = 100.0 * g(d, "SQ_WAIT_ANY") / (g(d, "SQ_WAIT_ANY") + g(d, "SQ_WAIT_INST_ANY") + g(d, "SQ_ACTIVE_INST_ANY")) if d.get("SQ_ACTIVE_INST_ANY") else float("nan")
    stall = 100.0 * g(d, "SQ_WAIT_INST_ANY") / (g(d, "SQ_WAIT_ANY") + g(d, "SQ_WAIT_INST_ANY") + g(d, "SQ_ACTIVE_INST_ANY")) if d.get("SQ_ACTIVE_INST_ANY") else float("nan")
    ldsst = 100.0 * g(d, "SQ_WAIT_INST_LDS") / g(d, "SQ_WAIT_INST_ANY") if d.get("SQ_WAIT_INST_ANY") else float("nan")
    conf = 100.0 * g(d, "SQ_LDS_BANK_CONFLICT") / g(d, "SQ_LDS_IDX_ACTIVE") if d.get("SQ_LDS_IDX_ACTIVE") else float("nan")
    mix = "%.0f : %.0f : %.0f : %.0f : %.0f" % tuple(g(d, c) / waves if waves == waves and waves else float("nan")
                                                    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM"))
    # SQ_VALU_MFMA_BUSY_CYCLES: cycles, summed over the SIMDs (32 per v_mfma_i32_32x32x32_i8); SQ_BUSY_CYCLES: summed over the 32 shader
    # engines, each with 32 SIMDs.  Cross-check at 10 000 samples: 515 M records / 32 per MFMA step x 4 MFMAs x 32 cycles = 2.06 G
    # SIMD-cycles = 0.84 ms of the 1024 SIMDs at 2.4 GHz = 14.7 % of k2_sorted_kernel's 5.7 ms; this column says 14.8 %.
    mfma = 100.0 * g(d, "SQ_VALU_MFMA_BUSY_CYCLES") / (32.0 * g(d, "SQ_BUSY_CYCLES")) if d.get("SQ_BUSY_CYCLES") and d.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None else float("nan")
    m = d["_meta"]
    lines.append("| `%s` | %.0f | %s+%s / %s / %s | %.1f | %.1f | %.1f | %.1f | %.1f | %.1f | %s | %.1f | %.1f | %.1f |" % (
        k, waves, m.get("vgpr"), m.get("accum_vgpr"), m.get("lds_bytes"), m.get("scratch_bytes"), lanes, valu_busy, wait, stall, ldsst, conf, mix, mfma,
        g(d, "FETCH_SIZE") / 1024.0, g(d, "WRITE_SIZE") / 1024.0))
open(os.path.join(out, tag + "_counters.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
