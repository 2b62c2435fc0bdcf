#!/usr/bin/env python3
"""Turn the rocprofv3 outputs of collect_profiles.sh into the small files committed under profiles/:
<tag>_kernel_stats.csv (our kernels only) and <tag>_traffic.json (HBM bytes per pass per kernel from the PMC
counters FETCH_SIZE / WRITE_SIZE, which rocprofv3 reports in KB)."""
import csv, json, os, sys
from collections import defaultdict

tag, fetch_csv, write_csv, out = sys.argv[1:5]


def short(name):
    for k in ("b3_decode_kernel", "b3_narrow_kernel", "b3_emit_kernel", "b2_apply_kernel", "b2_emit_kernel"):
        if k in name:
            return k
    return None


def count_mode(name):
    """upload-time instantiations: decode<true, *>, emit<*, false, *>, narrow<false>"""
    return ("b3_decode_kernel<true" in name or "b3_narrow_kernel<false>" in name or
            ("b3_emit_kernel<" in name and name.split("b3_emit_kernel<")[1].split(",")[1].strip() == "false") or
            ("b2_emit_kernel<false>" in name))


def per_kernel(path):
    """bytes per kernel over the dispatches of the timed passes only: everything after the last upload-time
    (count mode) dispatch"""
    tot, cnt = defaultdict(float), defaultdict(int)
    if not os.path.exists(path):
        return tot, cnt
    rows = list(csv.DictReader(open(path)))
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else None
    if key:
        rows.sort(key=lambda r: int(r[key]))
    last_upload = -1
    for i, row in enumerate(rows):
        if count_mode(row.get("Kernel_Name", "")):
            last_upload = i
    for row in rows[last_upload + 1:]:
        k = short(row.get("Kernel_Name", ""))
        if k is None:
            continue
        tot[k] += float(row["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


PASSES = 3      # --steps 2 --warmup 1
fetch, nf = per_kernel(fetch_csv)
write, nw = per_kernel(write_csv)
res = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs (--kernel-trace --kernel-include-regex 'b3_|b2_'), "
                 "python bench.py --no-cpu-baseline --steps 2 --warmup 1, mean over the %d passes; counters are KB, x1024; FETCH_SIZE can "
                 "under-count wide coalesced reads by up to 2x on gfx950 (MI355X_MICROARCH.md, HBM section), so this is a lower bound" % PASSES,
       "fetch_bytes": {k: v * 1024 / PASSES for k, v in fetch.items()},
       "write_bytes": {k: v * 1024 / PASSES for k, v in write.items()},
       "dispatch_rows": {"fetch": dict(nf), "write": dict(nw)}}
res["traffic_bytes_per_pass"] = sum(res["fetch_bytes"].values()) + sum(res["write_bytes"].values())
json.dump(res, open(os.path.join(out, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(res, indent=1))

allstats = os.path.join(out, tag + "_kernel_stats_all.csv")
if os.path.exists(allstats):
    with open(allstats) as f, open(os.path.join(out, tag + "_kernel_stats.csv"), "w") as g:
        for i, line in enumerate(f):
            if i == 0 or short(line):
                g.write(line)
    os.remove(allstats)
