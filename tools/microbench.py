#!/usr/bin/env python3
"""Kernel micro-benchmarks at the 4M-B bench shapes (R = 256*128 rows, D = 768): TFLOP/s or GB/s per kernel,
interleaved A/B of the tile configurations.  Run on the GPU box:  python tools/microbench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.hip import ops, _lib as L  # noqa: E402

dev = "cuda"
R, D, Hd = 256 * 128, 768, 2048


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def report(name, sec, flops=0, nbytes=0):
    extra = f"{flops / sec / 1e12:8.1f} TF/s" if flops else f"{nbytes / sec / 1e9:8.1f} GB/s"
    print(f"{name:44s} {sec * 1e6:9.1f} us  {extra}", flush=True)


x = rnd(R, D)
for cfg in (9 + 256,):
    L.lib.fm_set_gemm_nt_config(cfg)
    for name, N, K in (("qkv", 3 * D, D), ("proj", D, D), ("dX fc13 (K=4096)", D, 2 * Hd)):
        w, xin = rnd(N, K), rnd(R, K)
        out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
        report(f"nt cfg{cfg} bf16 {name} N={N} K={K}", timeit(lambda: ops.gemm_nt(xin, w, out)), 2.0 * R * N * K)
    w2, h = rnd(D, Hd), rnd(R, Hd)
    res = torch.zeros(R, D, device=dev)
    report(f"nt cfg{cfg} residual fc2 N={D} K={Hd}", timeit(lambda: ops.gemm_nt(h, w2, res, epilogue=L.EPI_RESIDUAL, res=res)), 2.0 * R * D * Hd)
    w1, w3 = rnd(Hd, D), rnd(Hd, D)
    gu, act = torch.empty(R, 2 * Hd, device=dev, dtype=torch.bfloat16), torch.empty(R, Hd, device=dev, dtype=torch.bfloat16)
    report(f"nt cfg{cfg} swiglu fc1|fc3 N=2x{Hd} K={D}", timeit(lambda: ops.gemm_nt(x, w1, act, epilogue=L.EPI_SWIGLU, w2=w3, out2=gu, Hp=Hd)), 4.0 * R * Hd * D)
L.lib.fm_set_gemm_nt_config(9 + 256)

for name, N, K in (("dW qkv", 3 * D, D), ("dW proj", D, D), ("dW fc2", D, Hd), ("dW fc1", Hd, D)):
    a_, b_ = rnd(R, N), rnd(R, K)
    out = torch.zeros(N, K, device=dev)
    for cfg in (0, 1):
        L.lib.fm_set_gemm_tn_config(cfg)
        report(f"tn cfg{cfg} {name} N={N} K={K}", timeit(lambda: ops.gemm_tn(a_, b_, out)), 2.0 * R * N * K)
    L.lib.fm_set_gemm_tn_config(1)

B, H, N = 256, 12, 128
qkv = rnd(B * N, 3 * D)
o = torch.empty(B * N, D, device=dev, dtype=torch.bfloat16)
sm, sl = torch.zeros(B, H, N, device=dev), torch.zeros(B, H, N, device=dev)
kp = (torch.rand(B, N, device=dev) < 0.1)
cs = torch.randint(0, 3, (B, N), device=dev).cumsum(-1).int()
mod = torch.randint(0, 7, (B, N), device=dev).sort(-1).values.short()
do = rnd(B * N, D)
dqkv = torch.empty(B * N, 3 * D, device=dev, dtype=torch.bfloat16)
for mname, kw in (("none", dict()), ("keypad", dict(mask_kind=L.MASK_KEYPAD, kpad=kp)),
                  ("decoder", dict(mask_kind=L.MASK_DECODER, cs=cs, modq=mod, modk=mod))):
    f = lambda: ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, N, N, 0.125, stat_m=sm, stat_l=sl, **kw)
    report(f"attn fwd {mname}", timeit(f), 4.0 * B * H * N * N * 64)
    g = lambda: ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, do, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                             B, H, N, N, 0.125, sm, sl, **kw)
    report(f"attn bwd {mname}", timeit(g), 10.0 * B * H * N * N * 64)

xf = torch.randn(R, D, device=dev)
w = torch.ones(D, device=dev)
y = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
mu, rs = torch.zeros(R, device=dev), torch.zeros(R, device=dev)
report("layernorm fwd", timeit(lambda: ops.layernorm_fwd(xf, w, None, y, mu, rs)), nbytes=R * D * 6)
dy, dx, dxb, dw = rnd(R, D), torch.zeros(R, D, device=dev), torch.empty(R, D, device=dev, dtype=torch.bfloat16), torch.zeros(D, device=dev)
report("layernorm bwd (+dres, +bf16 copy, +dw)", timeit(lambda: ops.layernorm_bwd(dy, xf, w, mu, rs, dx, dres=dx, dx_bf16=dxb, dw=dw)), nbytes=R * D * 16)
da, gu2, dgu = rnd(R, Hd), rnd(R, 2 * Hd), torch.empty(R, 2 * Hd, device=dev, dtype=torch.bfloat16)
report("swiglu bwd", timeit(lambda: ops.swiglu_bwd(da, gu2, dgu, Hd, Hd)), nbytes=R * Hd * 10)
