#!/usr/bin/env python3
"""Post-process a ``rocprofv3 --kernel-trace`` CSV of tools/overlap_dp.py: where do RCCL's kernels sit relative to the compute kernels?
python tools/overlap_trace.py <kernel_trace.csv>"""
import csv
import sys


def main(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    is_rccl = lambda n: "nccl" in n.lower() or "rccl" in n.lower()
    rccl = [r for r in rows if is_rccl(r[2])]
    comp = [r for r in rows if not is_rccl(r[2])]
    print(f"{len(rows)} kernel dispatches, {len(rccl)} of them RCCL")
    if not rccl:
        print("no RCCL kernel in the trace (world size 1: the collectives were elided before launch)")
        return
    # overlap of every RCCL kernel with the union of compute kernels
    ev = sorted((s, e) for s, e, _ in comp)
    merged = []
    for s, e in ev:
        if merged and s <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], e)
        else:
            merged.append([s, e])
    tot = under = 0
    import bisect
    starts = [m[0] for m in merged]
    for s, e, _ in rccl:
        tot += e - s
        i = max(0, bisect.bisect_right(starts, s) - 1)
        while i < len(merged) and merged[i][0] < e:
            under += max(0, min(e, merged[i][1]) - max(s, merged[i][0]))
            i += 1
    names = {}
    for s, e, n in rccl:
        d = names.setdefault(n[:60], [0, 0]); d[0] += 1; d[1] += e - s
    for n, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1])[:6]:
        print(f"  {c:5d} x {t / c / 1e3:8.1f} us  {n}")
    print(f"RCCL kernel time {tot / 1e6:.2f} ms, of which {under / 1e6:.2f} ms ({100.0 * under / max(tot, 1):.0f} %) ran while a compute kernel was running")
    # queueing: time between a RCCL kernel's start and the end of the compute kernel that was running when it started
    waits = []
    for s, e, _ in rccl:
        i = bisect.bisect_right(starts, s) - 1
        if i >= 0 and merged[i][1] > s:
            waits.append((merged[i][1] - s) / 1e3)
    if waits:
        waits.sort()
        print(f"RCCL kernels that started under a compute kernel: {len(waits)}, median remaining compute {waits[len(waits) // 2]:.0f} us")


if __name__ == "__main__":
    main(sys.argv[1])
