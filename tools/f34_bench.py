#!/usr/bin/env python3
"""Timings of the round-3 "next" rows on one MI355X (for rocprofv3 --kernel-trace --stats and stand-alone):
  f3  DeviceUnifiedMasking over a 4M-B mod7 batch of 256 (Dirichlet budgets, image masks, span masking of caption / det)
  f4  one VQ-VAE training step (ViT-B/16 encoder + ViT-B decoder, 224^2, 16384 x 32 codebook, batch 64): forward + hand-written backward"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402


def timed(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from bench import build_model
    from fourm.data.synthetic import device_masked_batch, device_masking_for
    from fourm.vq import VQVAE
    dev = "cuda"
    out = {}
    model = build_model("fm_base_12e_12d_swiglu_nobias", dev, "mod7")
    um = device_masking_for(model, 128, 128, device=dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    out["masking_ms_per_batch_256"] = timed(lambda: device_masked_batch(model, um, 256, device=dev, generator=gen), 10)
    um.max_tries = 8                     # fewer budget retries drawn in advance (upstream's default is 100 tries)
    out["masking_ms_per_batch_256_8_tries"] = timed(lambda: device_masked_batch(model, um, 256, device=dev, generator=gen), 10)
    del model
    vq = VQVAE(image_size=224, enc_type="vit_b_enc", dec_type="vit_b_dec", patch_size=16, post_mlp=True, codebook_size=16384, latent_dim=32,
               norm_codes=True, sync_codebook=False, threshold_ema_dead_code=0).to(dev).train()
    opt = torch.optim.AdamW(vq.parameters(), lr=1e-4)
    x = torch.rand(64, 3, 224, 224, device=dev) * 2 - 1

    def step():
        dec, cl = vq(x)
        (F.mse_loss(dec, x) + cl.sum()).backward()
        opt.step(); opt.zero_grad(set_to_none=True)
    ms = timed(step, 6)
    out["vqvae_train_step_ms_batch_64"] = ms
    out["vqvae_train_images_per_s"] = 64 / ms * 1e3
    # ViT-B enc + dec: 2 x 36.8 GFLOP forward per image, x 3 for the train step
    out["vqvae_train_mfu"] = 64 * 2 * 36.8e9 * 3 / (ms * 1e-3) / 2.5e15
    with torch.no_grad():
        vq.eval()
        tok = vq.tokenize(x)
        out["vqvae_decode_tokens_ms_batch_64"] = timed(lambda: vq.decode_tokens(tok), 6)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
