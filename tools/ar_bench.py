#!/usr/bin/env python3
"""Autoregressive caption decoding with the K/V cache on 4M-B mod7 (random weights): ms per generated token.
Run on the GPU box:  python tools/ar_bench.py [batch] [tokens]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from fourm.models.generate import GenerationSampler  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
from fourm.data.synthetic import synthetic_batch  # noqa: E402
model = bench.build_model("fm_base_12e_12d_swiglu_nobias", dev, "mod7").eval()
md = synthetic_batch(model, B, 128, 128, device=dev, seed=0)
cap = md["caption"]
for d in md.values():
    d["target_mask"][:] = True
cap["input_mask"][:] = True; cap["input_mask"][:, :2] = False
cap["target_mask"][:, 2:2 + T] = False
smp = GenerationSampler(model)
for rep in range(6):
    graphs = rep >= 3
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = smp.autoregressive_generate(md, "caption", temperature=1.0, top_k=50, top_p=0.0, use_eos=False, use_graphs=graphs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{'graphs' if graphs else 'eager '} batch {B}: {out.shape[1] - 1} tokens in {dt * 1e3:.1f} ms = {dt * 1e3 / (out.shape[1] - 1):.2f} ms/token, {B * (out.shape[1] - 1) / dt:.0f} tokens/s", flush=True)
