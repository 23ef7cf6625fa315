#!/usr/bin/env python3
"""Attention forward / backward alone at the 4M-B bench shape (B = 256, H = 12, 128 x 128 tokens), every mask kind.
FOURM_ATTN_BWD_DS=0 selects the two-softmax backward for comparison.  Run on the GPU box:  python tools/attn_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.hip import ops, _lib as L  # noqa: E402

dev = "cuda"


def timeit(fn, iters=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / iters * 1e-3)
    return best


B, H, N, D = 256, 12, int(os.environ.get("ATTN_N", 128)), 768
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
qkv, do = rnd(B * N, 3 * D), rnd(B * N, D)
o = torch.empty(B * N, D, device=dev, dtype=torch.bfloat16)
sm, sl = torch.zeros(B, H, N, device=dev), torch.zeros(B, H, N, device=dev)
kp = (torch.rand(B, N, device=dev) < 0.1)
cs = torch.randint(0, 3, (B, N), device=dev).cumsum(-1).int()
mod = torch.randint(0, 7, (B, N), device=dev).sort(-1).values.short()
dqkv = torch.empty(B * N, 3 * D, device=dev, dtype=torch.bfloat16)
for mname, kw in (("none", dict()), ("keypad", dict(mask_kind=L.MASK_KEYPAD, kpad=kp)),
                  ("decoder", dict(mask_kind=L.MASK_DECODER, cs=cs, modq=mod, modk=mod))):
    f = lambda: ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, N, N, 0.125, stat_m=sm, stat_l=sl, **kw)
    g = lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, do, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                             B, H, N, N, 0.125, sm, sl, **kw)
    tf, tb = timeit(f), timeit(g)
    print(f"{mname:8s} fwd {tf * 1e6:7.1f} us {4.0 * B * H * N * N * 64 / tf / 1e12:6.1f} TF/s   bwd {tb * 1e6:7.1f} us {10.0 * B * H * N * N * 64 / tb / 1e12:6.1f} TF/s", flush=True)
