#!/usr/bin/env python3
"""Long race screen of the staggered (ping-pong, persistent) GEMM schedules at the 4M-B shapes: every NT configuration
accumulates in the same k order, so each repetition must be BIT-identical to the lock-step configuration 2; the TN
schedule (fp32 atomics: order varies) must stay within 1e-5 of an fp64-free reference.  python tools/race_screen.py [--reps 200]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
from fourm.hip import ops, _lib as L

ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
dev = "cuda"
R = 256 * 128
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
bad = 0
for N, K in ((2304, 768), (768, 768), (768, 4096), (2048, 768)):
    x, w = rnd(R, K), rnd(N, K)
    out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
    L.lib.fm_set_gemm_nt_config(2 + 256)
    ops.gemm_nt(x, w, out); ref = out.clone()
    res0 = torch.randn(R, N, device=dev)
    outf = torch.empty(R, N, device=dev)
    ops.gemm_nt(x, w, outf, epilogue=L.EPI_RESIDUAL, res=res0); ref_res = outf.clone()
    for cfg in (7, 8, 10, 11):
        L.lib.fm_set_gemm_nt_config(cfg + 256)
        n_bad = 0
        for _ in range(a.reps):
            out.zero_(); ops.gemm_nt(x, w, out)
            n_bad += int(not torch.equal(out, ref))
            outf.zero_(); ops.gemm_nt(x, w, outf, epilogue=L.EPI_RESIDUAL, res=res0)
            n_bad += int(not torch.equal(outf, ref_res))
        print(f"nt cfg{cfg} N={N} K={K}: {n_bad} mismatching runs of {2 * a.reps}", flush=True)
        bad += n_bad
L.lib.fm_set_gemm_nt_config(9 + 256)
for N, K in ((2304, 768), (768, 768), (768, 2048)):
    A, B = rnd(R, N), rnd(R, K)
    ref = A.float().t() @ B.float()
    n_bad = 0
    for _ in range(a.reps // 4):
        out = torch.zeros(N, K, device=dev)
        ops.gemm_tn(A, B, out)
        err = float((out - ref).norm() / ref.norm())
        n_bad += int(not err < 1e-5)
    print(f"tn N={N} K={K}: {n_bad} runs over tolerance of {a.reps // 4}", flush=True)
    bad += n_bad
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad})")
