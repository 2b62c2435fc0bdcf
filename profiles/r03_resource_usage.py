#!/usr/bin/env python3
"""Compiler resource usage of the engine's kernels (hipcc -Rpass-analysis=kernel-resource-usage, gfx950): VGPRs, AGPRs, scratch,
LDS, the occupancy the registers allow.  No GPU needed.   python profiles/r03_resource_usage.py > profiles/r03_resource_usage.md"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ("k0_decode", "k1n_kernel", "k1w_kernel", "wrun_anc", "k2_apply", "k2_sorted", "cs_hist", "cs_scatter", "rs_hist", "rs_scatter", "rs_rows", "rg_hist",
        "rg_scatter", "ct_hist", "ct_scatter", "wide_count", "wide_expand", "n2a_", "d2_", "row_nnz", "row_compact", "a2a_tile", "a2a_global", "a2a_direct")
print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | static LDS B/workgroup | waves/SIMD the registers allow |")
print("|---|---|---|---|---|---|---|")
for f in ("a2a_blocks", "engine", "new2all", "db2db", "a2a_v1"):
    out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-I" + ROOT + "/include",
                          "-I" + ROOT + "/kmer-db_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", ROOT + "/kmer-db_amd/csrc/" + f + ".hip", "-o", "/dev/null"],
                         capture_output=True, text=True).stderr
    for b in out.split("Function Name: ")[1:]:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        if not any(k in dem for k in KEEP) or "rocprim" in dem:
            continue
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]      # noqa: E731
        print("| `%s` (%s.hip) | %s | %s | %s | %s | %s | %s |" % (dem.split("(")[0][:60], f, g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                                     g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
