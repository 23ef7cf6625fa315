#!/usr/bin/env python3
"""BASELINE.json configs[4]: RGB VQ tokenizer (ViT-B/16 encoder, 224^2 -> 14x14 codes, 16384 x 32 cosine
codebook) encode+quantize throughput on one MI355X.  Prints one JSON line (images/s, codes/s, fraction
of the bf16 MFMA roofline; algorithmic work 37.0 GFLOP/image)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.vq import VQ  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)      # save_vq_tokens.py:387-388 default sub-batch
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
a = ap.parse_args()
torch.manual_seed(0)
model = VQ(image_size=224, enc_type="vit_b_enc", patch_size=16, post_mlp=True, codebook_size=16384, latent_dim=32, norm_codes=True,
           sync_codebook=False).cuda().eval()
def run(batch):
    x = torch.rand(batch, 3, 224, 224, device="cuda") * 2 - 1
    for _ in range(a.warmup):
        model.tokenize(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        model.tokenize(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.steps


dt = run(a.batch)
out = {"metric": "RGB VQ tokenizer encode+quantize", "value": a.batch / dt, "unit": "images/s", "codes_per_s": a.batch * 196 / dt,
       "ms_per_batch": dt * 1e3, "batch": a.batch, "dtype": "bf16 (ViT) + f32 (code search)", "data": "synthetic",
       "roofline": {"bound": "mfma", "achieved": 37.0e9 * a.batch / dt / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                    "frac": 37.0e9 * a.batch / dt / 2.5e15}}
if a.batch != 256:      # the reference sub-batch (64) half-fills the chip; the same call at 256 images for comparison
    dt2 = run(256)
    out["at_batch_256"] = {"images_per_s": 256 / dt2, "ms_per_batch": dt2 * 1e3, "tflops": 37.0e9 * 256 / dt2 / 1e12}
print(json.dumps(out))
