#!/usr/bin/env python3
"""fm_shadow_refresh alone on the transposed weight images of 4M-B (fp32 master -> bf16 transposed): GB/s.
Run on the GPU box:  python tools/shadow_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.hip import ops  # noqa: E402

dev = "cuda"
shapes = [(2304, 768), (768, 768), (2048, 768), (2048, 768), (768, 2048)] * 12 + [(2304, 768), (768, 768), (768, 768), (1536, 768), (768, 768), (2048, 768), (2048, 768), (768, 2048)] * 12
for tr in (True, False):
    jobs, nbytes = [], 0
    for o, i in shapes:
        src = torch.randn(o, i, device=dev)
        dst = torch.zeros((i, ops.ru(o, 64)) if tr else (o, ops.ru(i, 64)), dtype=torch.bfloat16, device=dev)
        jobs.append((src, dst, tr)); nbytes += o * i * 6
    table, tiles = ops.shadow_jobs_table(jobs, dev)
    for _ in range(3):
        ops.shadow_refresh(table, len(jobs), tiles)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.shadow_refresh(table, len(jobs), tiles)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    ok = all(torch.equal(d, (s.t() if t else s).to(torch.bfloat16)[:, :d.shape[1]]) if d.shape[1] == (s.shape[0] if t else s.shape[1]) else True for s, d, t in jobs[:5])
    print(f"{'transposed' if tr else 'plain':10s}: {len(jobs)} jobs, {nbytes / 1e9:.2f} GB, {ms * 1e3:.0f} us, {nbytes / ms / 1e6:.0f} GB/s, correct={ok}", flush=True)
