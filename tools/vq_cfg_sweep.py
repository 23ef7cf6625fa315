#!/usr/bin/env python3
"""RGB tokenizer encode + quantize (batch 64) under every fm_gemm_nt tile configuration: which one suits M = 12 544 rows."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.hip import _lib as L  # noqa: E402

from fourm.vq import VQ  # noqa: E402
vq = VQ(image_size=224, enc_type="vit_b_enc", patch_size=16, post_mlp=True, codebook_size=16384, latent_dim=32, norm_codes=True,
        sync_codebook=False).cuda().eval()
x = torch.randn(int(os.environ.get("VQ_BATCH", 64)), 3, 224, 224, device="cuda")
for cfg in (9, 2, 0, 6, 7, 8, 10, 11, 1):
    L.lib.fm_set_gemm_nt_config(cfg + 256)
    try:
        for _ in range(3):
            vq.encode(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10):
            vq.encode(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 10
        print(f"cfg {cfg:2d}: {dt * 1e3:7.3f} ms/batch  {x.shape[0] / dt:9.0f} images/s", flush=True)
    except Exception as e:
        print(f"cfg {cfg}: {type(e).__name__}: {e}")
L.lib.fm_set_gemm_nt_config(9 + 256)
