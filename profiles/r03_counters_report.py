#!/usr/bin/env python3
"""profiles/r03_counters.md: the per-kernel counters of <tag>_counters.json read together with the kernel durations of
<tag>_kernel_stats_all.csv (rocprofv3 --stats of the same command).  python profiles/r03_counters_report.py > profiles/r03_counters.md"""
import csv, json, os
HERE = os.path.dirname(os.path.abspath(__file__))
SIMDS, GHZ = 1024, 2.4          # 256 CUs x 4 SIMDs; nominal shader clock (profiled passes run a little lower)


def durations(tag):
    out = {}
    for base in (os.path.join(HERE, tag + "_kernel_stats.csv"), os.path.join(HERE, "..", "gpurun_out", tag + "_kernel_stats_all.csv")):
        if os.path.exists(base):
            for r in csv.DictReader(open(base)):
                n = r["Name"]
                for k in ("k0_decode_kernel<false>", "k0_decode_kernel<true>", "k1n_kernel", "k1w_kernel", "k2_apply_kernel", "k2_sorted_kernel", "cs_hist_kernel", "cs_scatter_kernel",
                          "rs_hist_kernel", "rs_scatter_kernel", "rg_hist_kernel", "rg_scatter_kernel", "ct_hist_kernel", "ct_scatter_kernel", "n2a_walk_kernel", "n2a_probe_kernel",
                          "row_nnz_kernel", "row_compact_kernel", "d2_emit_kernel", "d2_probe_kernel"):
                    if k in n and int(r["Calls"]) >= 5:
                        kk = k.replace("<false>", "<short>").replace("<true>", "<long>")
                        out.setdefault(kk, float(r["AverageNs"]) / 1e6)
    return out


def table(tag, title):
    d = json.load(open(os.path.join(HERE, tag + "_counters.json")))
    dur = durations(tag)
    print("### %s\n" % title)
    print("| kernel | ms | waves | VGPRs / LDS B per workgroup | VALU instructions: share of all VALU issue slots of the chip | lanes active per VALU instruction | "
          "wave time: waiting (s_waitcnt, barrier) / issue-stalled / issuing | LDS-stall share of the issue stalls | LDS bank-conflict cycles | instructions per wave VALU : SALU : LDS : VMEM | "
          "matrix cores busy | HBM fetch + write |")
    print("|" + "---|" * 12)
    for k in sorted(d, key=lambda x: -dur.get(x, 0)):
        x = d[k]
        if k not in dur or "SQ_WAVES" not in x:
            continue
        g = lambda c: x.get(c, float("nan"))      # noqa: E731
        w = g("SQ_WAVES")
        slots = SIMDS * GHZ * 1e9 / 4 * dur[k] * 1e-3
        tot = g("SQ_WAIT_ANY") + g("SQ_WAIT_INST_ANY") + g("SQ_ACTIVE_INST_ANY")
        m = x["_meta"]
        mf = 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (SIMDS * GHZ * 1e9 * dur[k] * 1e-3) if "SQ_VALU_MFMA_BUSY_CYCLES" in x else float("nan")
        fw = "%.2f + %.2f GB" % (g("FETCH_SIZE") / 1e6 * 1.024, g("WRITE_SIZE") / 1e6 * 1.024) if "FETCH_SIZE" in x else "—"
        print("| `%s` | %.2f | %d | %s / %s | %.0f %% | %.0f %% | %.0f / %.0f / %.0f %% | %.0f %% | %.0f %% | %.0f : %.0f : %.0f : %.0f | %s | %s |" % (
            k, dur[k], w, int(m.get("vgpr") or 0) * 2 if False else m.get("vgpr"), m.get("lds_bytes"), 100 * g("SQ_INSTS_VALU") / slots,
            100 * g("SQ_THREAD_CYCLES_VALU") / (64 * g("SQ_ACTIVE_INST_VALU")), 100 * g("SQ_WAIT_ANY") / tot, 100 * g("SQ_WAIT_INST_ANY") / tot, 100 * g("SQ_ACTIVE_INST_ANY") / tot,
            100 * g("SQ_WAIT_INST_LDS") / max(1.0, g("SQ_WAIT_INST_ANY")), 100 * g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE")),
            g("SQ_INSTS_VALU") / w, g("SQ_INSTS_SALU") / w, g("SQ_INSTS_LDS") / w, g("SQ_INSTS_VMEM") / w, ("%.0f %%" % mf) if mf == mf and mf > 0.5 else "—", fw))
    print()


print("""# Per-kernel hardware counters, round 3 (`bash profiles/collect_counters.sh`, `python profiles/r03_counters_report.py`)

Counters: three SQ groups of 8 (one `rocprofv3 --pmc` pass each, only the engine's kernels instrumented), `FETCH_SIZE`, `WRITE_SIZE`;
means per launch over the four calls of `bench.py --steps 2 --warmup 1`.  Durations: `rocprofv3 --kernel-trace --stats` of the same
command without counters.  "VALU share" = `SQ_INSTS_VALU` / (1024 SIMDs x duration x 2.4 GHz / 4): a wave64 VALU instruction
occupies its SIMD's VALU for one quad-cycle (`SQ_ACTIVE_INST_VALU` equals `SQ_INSTS_VALU` within 1 %).  Lanes = `SQ_THREAD_CYCLES_VALU`
/ (64 x `SQ_ACTIVE_INST_VALU`) (a plain copy kernel measures 97 %).  Wave time = `SQ_WAIT_ANY` (parked at s_waitcnt / a barrier) +
`SQ_WAIT_INST_ANY` (ready but not issued) + `SQ_ACTIVE_INST_ANY`.  VGPR / LDS columns as rocprofv3 reports them per dispatch (VGPRs in
allocation granules of the 512-entry file per SIMD; `r03_resource_usage.md` has the compiler's exact figures); `FETCH_SIZE` counts half
of wide coalesced reads on gfx950 (MI355X_MICROARCH.md).

Reading: **K0 and K1n are bound by VALU issue** (three quarters of every VALU slot of the chip for the kernel's whole duration, at
half to two thirds of the lanes active: divergent decode loops, per-lane summaries) — not by HBM (2-5 GB in 1.0-1.5 ms) and not by
latency.  **K1w** issues a third of the slots and its waves are parked two thirds of their time: LDS round trips inside the per-batch
phases (walks, the binary search, LDS-atomic reservations) at 10-14 waves per CU; since the runs are handed out dynamically its time
is the sum of that, no longer the slowest wave.  **The sort kernels** (`cs_scatter`, `rs_scatter`, `rs_hist`) wait for memory: 3-8 % VALU,
60-90 % of the wave time parked; `rs_scatter` moves 5.6 (x2) + 14 GB in 7.1 ms = 3.5 TB/s.  **K2** (`k2_sorted`, `k2_apply`): VALU 50-60 %
of the slots (two bit-matrix transposes and the operand spreading per 64-record step), matrix cores 12-13 % busy, a third of the
LDS cycles lost to bank conflicts in the byte-spreading table reads.
""")
table("r03_v1_c2", "C2: 1000 x 5 Mbp, few streams (210): wide records through the arrival-order pool and one counting-sort pass")
table("r03_v1_c3part", "10 000 x 300 kbp, many streams (20 100): wide records into per-block-row chunks, one sort pass inside the rows")
table("r03_v1_mode_all2all-sp", "all2all-sp: 20 000 x 100 kbp, k = 25, f = 0.1 (row mode + CSR compaction)")
table("r03_v1_mode_new2all", "new2all: 1000 queries against 10 000 samples")
