#!/usr/bin/env python3
"""Kernel-trace view of `tools/coresidency_probe.py --trace`: per round (one dW list on stream A, four swiglu_bwd launches on stream B from a common
start) every kernel's interval relative to the round's start, and the round's wall time against the serial sum of the stand-alone durations.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/cores_trace -o t -- python $REPO/tools/coresidency_probe.py --trace
    python tools/coresidency_trace.py [csv] > profiles/r04_coresidency_trace.txt"""
import csv
import glob
import sys

f = sys.argv[1] if len(sys.argv) > 1 else glob.glob("gpurun_out/cores_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f, newline="")) if "gemm_tn_multi" in r["Kernel_Name"] or "swiglu_bwd" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
iv = [("G" if "gemm_tn_multi" in r["Kernel_Name"] else "S", int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
       "128x256 K64, 192 VGPR" if "128, 64" in r["Kernel_Name"] else "256x256 K32, 256 VGPR") for r in rows]
# rounds are separated by a host synchronisation: a kernel that starts >= 40 us after everything before it has ended opens a new round
rounds, cur, hi = [], [], 0
for x in iv:
    if cur and x[1] - hi >= 40_000:
        rounds.append(cur); cur = []
    cur.append(x); hi = max(hi, x[2])
rounds.append(cur)
rounds = [r for r in rounds if sum(1 for x in r if x[0] == "G") == 1 and sum(1 for x in r if x[0] == "S") == 4]
alone_s = min(e - s for k, s, e, _ in iv if k == "S")
print("# rocprofv3 --kernel-trace of tools/coresidency_probe.py --trace: one dW list of a 4M-B encoder layer (stream A) and 4 x swiglu_bwd (stream B) per round,")
print("# issued from a common start; [start, end] in us relative to the round's first kernel.  Stand-alone: swiglu_bwd %.0f us, the dW list ~405 us (256 x 256 form) /" % (alone_s / 1e3))
print("# ~505 us (128 x 256 form) (profiles/r04_coresidency_probe.txt).")
for r in rounds:
    t0 = min(x[1] for x in r)
    g = [x for x in r if x[0] == "G"][0]
    ss = [x for x in r if x[0] == "S"]
    wall = max(x[2] for x in r) - t0
    print(f"{g[3]:22s} | dW [{(g[1] - t0) / 1e3:6.1f}, {(g[2] - t0) / 1e3:6.1f}] = {(g[2] - g[1]) / 1e3:6.1f} us | swiglu_bwd " +
          "  ".join(f"[{(a - t0) / 1e3:6.1f}, {(b - t0) / 1e3:6.1f}]" for _, a, b, _ in ss) + f" | round wall {wall / 1e3:6.1f} us")
print("""# Reading.  Default form (a workgroup owns its CU): the streaming launch that is running when the GEMM starts finishes; the NEXT one is admitted only as
# GEMM workgroups retire - its interval spans the whole GEMM (486 - 573 us for 124 us of work) - and the last two run alone: the two streams serialise
# (round wall ~ 850 - 950 us against a serial sum of 405 + 4 x 124 = 900 us).  128 x 256 form (128 registers per lane and SIMD left): the kernels ARE resident together -
# and both slow down: the dW list takes ~850 us instead of ~505, each co-running swiglu_bwd 300 - 314 us instead of 124; round wall ~ 990 us = the
# serial sum of the stand-alone times (505 + 4 x 124 = 1000 us).  Co-residency buys nothing: alone the dW list pulls ~3.3 TB/s through the fabric and swiglu_bwd 5.4 TB/s
# from HBM; together they share one memory system that delivers ~6 TB/s, at two streaming waves per SIMD.""")
