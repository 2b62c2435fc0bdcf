#!/bin/bash
# Round 5, job 11: the default flow (C2, then c3part, then c3gpu in one process) — job 10's line had c3part's wide stage at 27 ms there (7.7 standalone)
TAG=r05_j11
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
KMDB_VERBOSE=1 timeout 900 python bench.py --no-cpu-baseline 2> $OUT/${TAG}_bench.err > $OUT/${TAG}_bench.json
grep -E "wide kernel:|sized too small|enlarged|doubling|slices of" $OUT/${TAG}_bench.err | cut -c1-250
python - <<PY
import json
b=json.load(open("$OUT/${TAG}_bench.json"))
print("c2", round(b["ms_per_step"],3), b["roofline"]["per_kernel_ms"])
for n in ("c3part","c3gpu"):
    e=b["extra"][n]; print(n, round(e["ms_per_step"],3), e["step_kernel_ms"], e["per_kernel_ms"])
PY
rm -rf /tmp/prof_def
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_def -- python $OLDPWD/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $OUT/${TAG}_bench_traced.json 2> /tmp/prof_def.err )
python - <<PY
import csv, glob, collections
best=None
for f in glob.glob("/tmp/prof_def/**/*kernel_trace.csv", recursive=True):
    rows=list(csv.DictReader(open(f)))
    if any("k1w_kernel" in r["Kernel_Name"] for r in rows): best=rows
if best:
    d=collections.defaultdict(list)
    for r in best:
        n=r["Kernel_Name"]
        for k in ("k1w_kernel","l2_join_apply","l2_lists","k1n_kernel","rs_scatter","k2_jobs","k0_decode_kernel<true>","k0_decode_kernel<false>","wide_expand","wrun_anc","l2_ranks"):
            if k in n: d[k].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    for k,v in d.items(): print(k, [round(x,2) for x in v])
PY
