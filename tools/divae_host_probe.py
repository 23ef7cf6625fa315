#!/usr/bin/env python3
"""Is a UNet evaluation of the DiVAE detokenizer bound by the host's enqueue rate?  Enqueue time of N evaluations (no synchronisation) against
their total time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
from fourm.vq import DiVAE
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = DiVAE(image_size=224, n_channels=3, enc_type="vit_b_enc", patch_size=16, codebook_size=16384, latent_dim=32, post_mlp=True, norm_codes=True,
          scheduler="ddim", prediction_type="sample", beta_schedule="linear", sync_codebook=False)
for p in m.decoder.parameters():
    if float(p.detach().abs().max()) == 0:
        torch.nn.init.normal_(p, std=0.02)
m = m.cuda().eval()
tokens = torch.randint(0, 16384, (B, 14, 14), device="cuda")
quant = m.tokens_to_embedding(tokens).float()
x = torch.randn(B, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m.decoder(x, 500, quant)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        m.decoder(x, 500, quant)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print(f"batch {B}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms per evaluation, total {1e3 * (t2 - t0) / n:.2f} ms per evaluation")
