#!/usr/bin/env python3
"""Kernel statistics (calls, total / average duration, share) from a rocprofv3 rocpd SQLite trace
(`rocprofv3 --kernel-trace --stats -d DIR -o trace -- <cmd>` writes DIR/trace_results.db).
    python tools/rocpd_stats.py gpurun_out/prof/trace_results.db > profiles/<name>_kernel_stats.csv"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                   "group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
for r in rows:
    w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100.0 * r[2] / tot:.3f}", r[4], r[5]])
