#!/usr/bin/env python3
"""fm_gemm_nt time vs reduction length at fixed M, N: T(K) = fixed + slope * K separates the per-tile prologue /
epilogue cost from the main-loop rate.  python tools/k_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
from fourm.hip import ops, _lib as L

dev = "cuda"
R = 256 * 128
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


for N in (2304, 768):
    for cfg in (2, 7, 1):
        L.lib.fm_set_gemm_nt_config(cfg + 256)
        row = []
        for K in (64, 256, 512, 768, 1536, 3072, 6144):
            x, w = rnd(R, K), rnd(N, K)
            out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
            t = timeit(lambda: ops.gemm_nt(x, w, out))
            row.append((K, t))
        (k0, t0), (k1, t1) = row[3], row[-1]
        slope = (t1 - t0) / (k1 - k0)
        fixed = t0 - slope * k0
        print(f"N={N} cfg{cfg}: " + "  ".join(f"K{k}:{t:7.1f}us" for k, t in row) + f" | fixed {fixed:6.1f} us, slope {slope * 64:6.2f} us per 64 of K "
              f"(= {2.0 * R * N * 64 / (slope * 64) / 1e6:6.0f} TF/s main loop)", flush=True)
L.lib.fm_set_gemm_nt_config(9 + 256)
