#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of collect_profiles.sh into the small files committed under profiles/:
<tag>_kernel_stats.csv (the kernels of the all2all call only) and <tag>_traffic.json (HBM bytes per call per kernel from the PMC
counters FETCH_SIZE / WRITE_SIZE, which rocprofv3 reports in KB)."""
import csv, json, os, sys
from collections import defaultdict

tag, fetch_csv, write_csv, out = sys.argv[1:5]
CALLS = 1 + 1 + 2          # bench.py --steps 2 --warmup 1: the cold call, one warm-up, two timed calls
OURS = ("k0_decode_kernel", "k0_short_direct_kernel", "l2_ranks_kernel", "l2_offsets_kernel", "l2_lists_kernel", "l2_join_apply_kernel", "k2j_starts_kernel", "k2j_starts_flat_kernel", "k2j_build_kernel", "k2_jobs_kernel", "k2d_kernel", "k1n_kernel", "k1w_kernel", "wrun_anc_kernel", "ct_hist_kernel", "ct_scatter_kernel", "ct_count_kernel", "rs_rows_kernel", "rs_hist_kernel", "rs_scatter_kernel", "k1g_kernel", "k2_apply_kernel", "k2_sorted_kernel", "cs_hist_kernel", "cs_scatter_kernel", "csl_scatter_kernel", "wide_count_kernel", "wide_expand_kernel",
        "count_chunks_kernel", "fill_u32_kernel", "rg_hist_kernel", "rg_scatter_kernel", "n2a_", "d2_", "row_nnz_kernel", "row_compact_kernel")


def short(name):
    for k in OURS:
        if k in name:
            return k
    if "rocprim" in name and ("radix" in name or "onesweep" in name):
        return "rocprim radix sort (chunk table, wide records)"
    if "rocprim" in name and ("scan" in name or "reduce" in name):
        return "rocprim scan / reduce"
    if "fillBuffer" in name:
        return "memset (matrix, chunk table, wide-pool keys, cursors)"
    if "cs_hist_kernel" in name or "cs_scatter_kernel" in name:
        return "counting sort of the wide pool"
    return None


def per_kernel(path):
    """KB per kernel over the dispatches of the all2all calls: everything after the last upload kernel (lay_* / width estimate)"""
    tot, cnt = defaultdict(float), defaultdict(int)
    if not os.path.exists(path):
        return tot, cnt
    rows = list(csv.DictReader(open(path)))
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    last_upload = -1
    for i, row in enumerate(rows):
        n = row.get("Kernel_Name", "")
        if "lay_" in n or "width_estimate" in n or "iota_u32" in n:
            last_upload = i
    for row in rows[last_upload + 1:]:
        k = short(row.get("Kernel_Name", ""))
        if k is None:
            continue
        tot[k] += float(row["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


stats_all = os.path.join(out, tag + "_kernel_stats_all.csv")
if os.path.exists(stats_all):
    rows = list(csv.DictReader(open(stats_all)))
    keep = [r for r in rows if any(k in r["Name"] for k in OURS + ("lay_", "width_estimate")) or
            ("rocprim" in r["Name"] and "at::" not in r["Name"] and 5 <= int(r["Calls"]) <= 64)]
    with open(os.path.join(out, tag + "_kernel_stats.csv"), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        wr.writeheader()
        for r in keep:
            r = dict(r)
            r["Name"] = r["Name"].replace("(anonymous namespace)::", "")[:160]
            wr.writerow(r)

fetch, nf = per_kernel(fetch_csv)
write, nw = per_kernel(write_csv)
if fetch or write:
    res = {"pipeline": "r02",
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (--kernel-trace), python bench.py --no-cpu-baseline "
                     "--steps 2 --warmup 1, mean over its %d all2all calls (the dispatches after the last upload kernel); counters are KB, "
                     "x1024; FETCH_SIZE can under-count wide coalesced reads by up to 2x on gfx950 (MI355X_MICROARCH.md, HBM section), so this "
                     "is a lower bound" % CALLS,
           "fetch_bytes": {k: v * 1024 / CALLS for k, v in fetch.items()},
           "write_bytes": {k: v * 1024 / CALLS for k, v in write.items()},
           "dispatch_rows": {"fetch": dict(nf), "write": dict(nw)}}
    res["traffic_bytes_per_pass"] = sum(res["fetch_bytes"].values()) + sum(res["write_bytes"].values())
    json.dump(res, open(os.path.join(out, tag + "_traffic.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))
