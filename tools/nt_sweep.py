#!/usr/bin/env python3
"""A/B of the fm_gemm_nt tile configurations at the 4M-B shapes (R = 256*128 rows) + a race screen: every
configuration accumulates in the same k order, so outputs must be BIT-identical to configuration 2's, run after run.
python tools/nt_sweep.py [--reps 3] [--screen 10]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
from fourm.hip import ops, _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3); ap.add_argument("--screen", type=int, default=10)
ap.add_argument("--cfgs", default="1,2,6,7,8,257,258,263,264")
a = ap.parse_args()
dev = "cuda"
R, D, Hd = 256 * 128, 768, 2048
cfgs = [int(c) for c in a.cfgs.split(",")]


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


cases = []
def case(name, flops, make):
    cases.append((name, flops, make))

def plain(N, K):
    w, x = rnd(N, K), rnd(R, K)
    out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
    return (lambda: ops.gemm_nt(x, w, out)), (lambda: out)
def resid(N, K):
    w, x = rnd(N, K), rnd(R, K)
    res = torch.randn(R, N, device=dev); out = torch.empty(R, N, device=dev)
    return (lambda: ops.gemm_nt(x, w, out, epilogue=L.EPI_RESIDUAL, res=res)), (lambda: out)
def swiglu():
    w1, w3, x = rnd(Hd, D), rnd(Hd, D), rnd(R, D)
    gu, act = torch.empty(R, 2 * Hd, device=dev, dtype=torch.bfloat16), torch.empty(R, Hd, device=dev, dtype=torch.bfloat16)
    return (lambda: ops.gemm_nt(x, w1, act, epilogue=L.EPI_SWIGLU, w2=w3, out2=gu, Hp=Hd)), (lambda: torch.cat([gu, act], 1))

case("qkv      N=2304 K=768 ", 2.0 * R * 2304 * 768, lambda: plain(2304, 768))
case("proj     N=768  K=768 ", 2.0 * R * 768 * 768, lambda: plain(768, 768))
case("dX fc2   N=2048 K=768 ", 2.0 * R * 2048 * 768, lambda: plain(2048, 768))
case("dX fc13  N=768  K=4096", 2.0 * R * 768 * 4096, lambda: plain(768, 4096))
case("fc2+res  N=768  K=2048", 2.0 * R * 768 * 2048, lambda: resid(768, 2048))
case("proj+res N=768  K=768 ", 2.0 * R * 768 * 768, lambda: resid(768, 768))
case("swiglu   N=2x2048 K=768", 4.0 * R * Hd * 768, swiglu)

for name, flops, make in cases:
    fn, get = make()
    L.lib.fm_set_gemm_nt_config(2)
    fn(); ref = get().clone()
    best = {c: 1e9 for c in cfgs}
    bad = {c: 0 for c in cfgs}
    for rep in range(a.reps):
        for c in cfgs:
            L.lib.fm_set_gemm_nt_config(c)
            best[c] = min(best[c], timeit(fn))
    for c in cfgs:
        L.lib.fm_set_gemm_nt_config(c)
        for _ in range(a.screen):
            fn()
            if not torch.equal(get(), ref):
                bad[c] += 1
    print(name + " | " + "  ".join(f"c{c}:{flops / best[c] / 1e12:6.0f}{'!' * min(bad[c], 3)}" for c in cfgs), flush=True)
L.lib.fm_set_gemm_nt_config(9 + 256)
