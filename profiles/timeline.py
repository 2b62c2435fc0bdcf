#!/usr/bin/env python3
"""Timeline of ONE all2all call from a rocprofv3 --kernel-trace CSV (the last call in the file): for every kernel its queue, start offset from
the call's first kernel, duration, and the idle time of its queue before it.  Usage: timeline.py <kernel_trace.csv> [out.md]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    if not rows:
        print("no rows")
        return
    name_k = next(k for k in rows[0] if k.lower() == "kernel_name")
    s_k = next(k for k in rows[0] if k.lower().startswith("start"))
    e_k = next(k for k in rows[0] if k.lower().startswith("end"))
    q_k = next((k for k in rows[0] if k.lower() == "queue_id"), None)
    ev = sorted(((int(r[s_k]), int(r[e_k]), r[name_k], r[q_k] if q_k else "0") for r in rows), key=lambda t: t[0])
    # the last call = from the last k0_decode_kernel<false> (walk back over the pool initialisation in front of it) to the end
    starts = [i for i, t in enumerate(ev) if "k0_decode_kernel<false>" in t[2] or ("k0_decode_kernel" in t[2] and "<0" in t[2])]
    if not starts:
        starts = [i for i, t in enumerate(ev) if "k0_decode_kernel" in t[2]]
    i0 = starts[-1] if len(starts) < 2 else starts[-1]
    while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 200_000 and not any(x in ev[i0 - 1][2] for x in ("k2_sorted", "k2_jobs", "k2_apply", "row_compact")):
        i0 -= 1
    call = ev[i0:]
    # the call ends with the copy of its counters after the last apply kernel: what follows (the caller's own kernels) is not part of it
    last = max((i for i, t in enumerate(call) if any(x in t[2] for x in ("k2_sorted", "k2_jobs", "k2_apply", "row_compact"))), default=len(call) - 1)
    end = last
    while end + 1 < len(call) and "rocclr_copyBuffer" in call[end + 1][2] and call[end + 1][0] - call[end][1] < 100_000:
        end += 1
    call = call[:end + 1]
    t0 = call[0][0]
    last_end = {}
    out = ["| kernel | queue | start (us) | duration (us) | queue idle before (us) |", "|---|---|---|---|---|"]
    busy = 0
    for s, e, n, q in call:
        idle = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = max(e, last_end.get(q, 0))
        short = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if len(short) > 70:
            short = short[:67] + "..."
        out.append("| `%s` | %s | %.1f | %.1f | %.1f |" % (short, q, (s - t0) / 1e3, (e - s) / 1e3, idle))
    span = (max(e for _, e, _, _ in call) - t0) / 1e3
    # time during which no kernel of the call runs at all
    iv = sorted((s, e) for s, e, _, _ in call)
    covered, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            covered += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    covered += cur_e - cur_s
    out.append("")
    out.append("call span %.1f us, some kernel running %.1f us, nothing running %.1f us (%d kernels)" % (span, covered / 1e3, span - covered / 1e3, len(call)))
    txt = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt[-1500:])


if __name__ == "__main__":
    main()
