#!/usr/bin/env python3
"""Diffusion detokenizer on one MI355X: one evaluation of the unet_patched decoder (196 M parameters, 224^2 images = 56 x 56 patch grid,
14 x 14 x 32 conditioning) and a DDIM decode of a batch of token grids.  python tools/divae_bench.py [batch] [ddim steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.vq import DiVAE  # noqa: E402
from fourm.hip import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
torch.manual_seed(0)
m = DiVAE(image_size=224, n_channels=3, enc_type="vit_b_enc", patch_size=16, codebook_size=16384, latent_dim=32, post_mlp=True, norm_codes=True,
          scheduler="ddim", prediction_type="sample", beta_schedule="linear", sync_codebook=False)
for p in m.decoder.parameters():          # (upstream zero-initialises the block tails: give every GEMM real operands)
    if float(p.abs().max()) == 0:
        torch.nn.init.normal_(p, std=0.02)
m = m.cuda().eval()
tokens = torch.randint(0, 16384, (B, 14, 14), device="cuda")
quant = m.tokens_to_embedding(tokens).float()
x = torch.randn(B, 3, 224, 224, device="cuda")
for _ in range(2):
    m.decoder(x, 500, quant)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    m.decoder(x, 500, quant)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
# algorithmic FLOPs of one evaluation: 3x3 / 1x1 convolutions + attention + embeddings, from the module tree
fl = 0.0
hw = {0: 56 * 56}
from fourm.vq.models.unet.unet import _Res, _Attn, _Down, _Up  # noqa: E402


class Prof:
    def __init__(self):
        self.recs = []

    def launch(self, name, flops, nbytes, tag=""):
        p = self

        class C:
            def __enter__(s):
                s.a, s.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.a.record()
                return s

            def __exit__(s, *e):
                s.b.record()
                p.recs.append((name, flops, s.a, s.b))
                return False
        return C()


prof = Prof()
ops.set_profiler(prof)
m.decoder(x, 500, quant)
ops.set_profiler(None)
torch.cuda.synchronize()
gemm_ms = sum(a.elapsed_time(b) for nme, f, a, b in prof.recs if nme.startswith("gemm_nt"))
gemm_fl = sum(f for nme, f, a, b in prof.recs if nme.startswith("gemm_nt"))
print(f"unet_patched, batch {B}: {dt * 1e3:.1f} ms per evaluation ({dt * 1e3 / B:.2f} ms per image); GEMMs {gemm_ms:.1f} ms = {gemm_fl / 1e9 / B:.0f} GFLOP per image at "
      f"{gemm_fl / gemm_ms / 1e9:.0f} TFLOP/s; everything else (im2col, GroupNorm, attention, adds) {dt * 1e3 - gemm_ms:.1f} ms")
t0 = time.perf_counter()
img = m.decode_tokens(tokens, timesteps=steps, generator=torch.Generator().manual_seed(0), verbose=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"decode_tokens, {steps} DDIM steps, batch {B}: {dt:.2f} s = {B / dt:.1f} images/s; output {tuple(img.shape)}, finite {bool(torch.isfinite(img).all())}")
