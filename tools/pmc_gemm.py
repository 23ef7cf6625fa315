#!/usr/bin/env python3
"""A handful of launches of the two GEMM kernels at bench shapes, for rocprofv3 --pmc runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.hip import ops, _lib as L  # noqa: E402

R, D, Hd = 256 * 128, 768, 2048
rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
x4, w4, o = rnd(R, 2 * Hd), rnd(D, 2 * Hd), torch.empty(R, D, device="cuda", dtype=torch.bfloat16)
xq, wq, oq = rnd(R, D), rnd(3 * D, D), torch.empty(R, 3 * D, device="cuda", dtype=torch.bfloat16)
a_, b_, dw = rnd(R, D), rnd(R, Hd), torch.zeros(D, Hd, device="cuda")
for _ in range(3):
    ops.gemm_nt(x4, w4, o)          # N=768  K=4096
    ops.gemm_nt(xq, wq, oq)         # N=2304 K=768
    ops.gemm_tn(a_, b_, dw)         # dW fc2
torch.cuda.synchronize()
