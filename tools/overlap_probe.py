#!/usr/bin/env python3
"""Do the dX chain (NT GEMMs, LayerNorm backward) and the weight-gradient GEMMs (TN) overlap usefully when they
are issued on two HIP streams?  Same launches, one stream vs two.  python tools/overlap_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
from fourm.hip import ops, _lib as L

dev = "cuda"
R, D, Hd = 256 * 128, 768, 2048
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
x, xh = rnd(R, D), rnd(R, Hd)
w_qkv, w_proj, w_fc2t, w13t = rnd(3 * D, D), rnd(D, D), rnd(Hd, D), rnd(D, 2 * Hd)
o_qkv, o_proj, o_h, o_d = (torch.empty(R, n, device=dev, dtype=torch.bfloat16) for n in (3 * D, D, Hd, D))
dgu = rnd(R, 2 * Hd)
g32, xf = torch.randn(R, D, device=dev), torch.randn(R, D, device=dev)
mu, rs, wln = torch.zeros(R, device=dev), torch.ones(R, device=dev), torch.ones(D, device=dev)
gbf = torch.empty(R, D, device=dev, dtype=torch.bfloat16)
dw_qkv, dw_proj, dw_fc1, dw_fc2 = (torch.zeros(a, b, device=dev) for a, b in ((3 * D, D), (D, D), (Hd, D), (D, Hd)))
dwln = torch.zeros(D, device=dev)


def chain():        # one encoder block's dX-side launches (shapes only; no data dependence needed for timing)
    ops.gemm_nt(x, w_fc2t, o_h, N=Hd, K=D)                      # dX fc2
    ops.gemm_nt(dgu, w13t, o_d, N=D, K=2 * Hd)                  # dX fc1|fc3
    ops.layernorm_bwd(o_d, xf, wln, mu, rs, g32, dres=g32, dx_bf16=gbf, dw=dwln)
    ops.gemm_nt(x, w_proj, o_proj, N=D, K=D)                    # dX proj
    ops.gemm_nt(o_qkv, rnd_w_qkv_t, o_d, N=D, K=3 * D)          # dX qkv
    ops.layernorm_bwd(o_d, xf, wln, mu, rs, g32, dres=g32, dx_bf16=gbf, dw=dwln)


def dws():
    ops.gemm_tn(x, xh, dw_fc2)                                  # dW fc2: N=768 K=2048
    ops.gemm_tn(dgu[:, :Hd], x, dw_fc1)                         # dW fc1
    ops.gemm_tn(dgu[:, Hd:], x, dw_fc1)                         # dW fc3
    ops.gemm_tn(x, x, dw_proj)                                  # dW proj
    ops.gemm_tn(o_qkv, x, dw_qkv)                               # dW qkv


rnd_w_qkv_t = rnd(D, 3 * D)
side = torch.cuda.Stream()


def timeit(fn, iters=12):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def serial():
    chain(); dws()


def two_streams():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dws()
    chain()
    main.wait_stream(side)


print(f"dX chain alone        {timeit(chain) * 1e3:8.1f} us")
print(f"dW GEMMs alone        {timeit(dws) * 1e3:8.1f} us")
print(f"one stream            {timeit(serial) * 1e3:8.1f} us")
print(f"two streams           {timeit(two_streams) * 1e3:8.1f} us")
