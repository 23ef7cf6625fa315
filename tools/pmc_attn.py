#!/usr/bin/env python3
"""A few launches of the attention kernels at bench shapes, for rocprofv3 --pmc runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.hip import ops, _lib as L  # noqa: E402

B, H, N, D = 256, 12, 128, 768
rnd = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
qkv, o, do, dqkv = rnd(B * N, 3 * D), torch.empty(B * N, D, device="cuda", dtype=torch.bfloat16), rnd(B * N, D), torch.empty(B * N, 3 * D, device="cuda", dtype=torch.bfloat16)
sm, sl = torch.zeros(B, H, N, device="cuda"), torch.zeros(B, H, N, device="cuda")
kp = torch.rand(B, N, device="cuda") < 0.1
for _ in range(3):
    ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, B, H, N, N, 0.125, stat_m=sm, stat_l=sl, mask_kind=L.MASK_KEYPAD, kpad=kp)
    ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], o, do, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], B, H, N, N, 0.125, sm, sl,
                 mask_kind=L.MASK_KEYPAD, kpad=kp)
torch.cuda.synchronize()
