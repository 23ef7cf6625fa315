#!/usr/bin/env python3
"""HBM-side traffic per launch of every kernel family of the train step, from two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE; separate passes, per MI355X_MICROARCH.md) over bench.py's --pmc-worker (two 4M-B steps at batch 256).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 reports half of a wide coalesced read; WRITE_SIZE uncalibrated).
python tools/traffic_table.py > gpurun_out/traffic_table.txt"""
import csv, glob, os, re, shutil, subprocess, sys, tempfile, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILIES = [
    ("gemm_nt plain (EPI 0)", r"gemm_nt_kernel<\d+, \d+, \d+, \d+, \d+, \d+, 0, false"),
    ("gemm_nt residual (EPI 2)", r"gemm_nt_kernel<\d+, \d+, \d+, \d+, \d+, \d+, 2, false"),
    ("gemm_nt SwiGLU (EPI 3)", r"gemm_nt_kernel<\d+, \d+, \d+, \d+, \d+, \d+, 3, false"),
    ("gemm_nt grouped (heads)", r"gemm_nt_kernel<\d+, \d+, \d+, \d+, \d+, \d+, 0, true"),
    ("gemm_tn_multi (dW of a layer)", r"gemm_tn_multi_kernel"), ("gemm_tn", r"gemm_tn_kernel<\w+, false"), ("gemm_tn grouped (heads)", r"gemm_tn_kernel<\w+, true"),
    ("attn_fwd", r"attn_fwd_kernel"), ("attn_bwd", r"attn_bwd_kernel"),
    ("ln_fwd", r"ln_fwd_kernel"), ("ln_bwd", r"ln_bwd_kernel"), ("swiglu_bwd", r"swiglu_bwd_kernel"),
    ("adamw + bf16 image", r"adamw_shadow_kernel"), ("adamw", r"adamw_kernel"), ("ce_fwd", r"ce_fwd_kernel"), ("ce_bwd", r"ce_bwd_kernel"),
    ("select_embed", r"select_embed_kernel"), ("embed_bwd", r"embed_bwd_kernel"), ("shadow_refresh", r"shadow_refresh_kernel"),
    ("sumsq", r"sumsq_kernel"),
]
kb = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.defaultdict(int)
dur = collections.defaultdict(float)
for counter in kb:
    d = tempfile.mkdtemp(prefix="fm_pmc_", dir="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-worker"]
    subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=400, capture_output=True)
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    for row in csv.DictReader(open(f, newline="")):
        if row["Counter_Name"] != counter:
            continue
        for name, rx in FAMILIES:
            if re.search(rx, row["Kernel_Name"]):
                kb[counter][name] += float(row["Counter_Value"])
                if counter == "FETCH_SIZE":
                    cnt[name] += 1
                    dur[name] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-3
                break
    shutil.rmtree(d, ignore_errors=True)
print(f"{'kernel family':30s} {'launches':>8s} {'read MB':>10s} {'write MB':>10s} {'total MB':>10s} {'us (under pmc)':>15s} {'GB/s':>9s}   per launch, 2 steps of 4M-B @ batch 256")
for name, _ in FAMILIES:
    n = cnt[name]
    if not n:
        continue
    rd, wr = 2 * kb["FETCH_SIZE"][name] * 1024 / n, kb["WRITE_SIZE"][name] * 1024 / n
    us = dur[name] / n
    print(f"{name:30s} {n:8d} {rd / 1e6:10.1f} {wr / 1e6:10.1f} {(rd + wr) / 1e6:10.1f} {us:15.1f} {(rd + wr) / us / 1e3:9.0f}")
