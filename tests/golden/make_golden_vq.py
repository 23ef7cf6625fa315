#!/usr/bin/env python3
"""Golden fixtures for the VQ tokenizer front end from the UNMODIFIED upstream ``fourm.vq.vqvae.VQ``
(container only).  Weights / images are regenerated from seeds on the test side; the fixture keeps the
upstream tokens, latents and quantised vectors.   python tests/golden/make_golden_vq.py [--check]"""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_stubs  # noqa: E402

ref_stubs.install()
from fourm.vq.vqvae import VQ as RefVQ, VQVAE as RefVQVAE  # noqa: E402

from oracle import vq_oracle as V  # noqa: E402
from tests.golden.cases import VQ_CASES, VQVAE_CASES  # noqa: E402


def run(name, check):
    c = VQ_CASES[name]
    cfg = V.vq_cfg(c["enc_type"], image=c["image"], patch=c["patch"], codebook=c["codebook"], post_mlp=c["post_mlp"])
    sd = V.seeded_vq_state_dict(cfg, seed=c["seed"])
    x = V.synthetic_images(cfg, c["batch"], seed=c["seed"])
    ref = RefVQ(image_size=cfg.image, enc_type=c["enc_type"], patch_size=cfg.patch, post_mlp=cfg.post_mlp, codebook_size=cfg.codebook,
                latent_dim=cfg.latent, norm_codes=True, sync_codebook=False)
    msg = ref.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    ref.eval()
    with torch.no_grad():
        h = ref.quant_proj(ref.encoder(x))
        quant, loss, tokens = ref.encode(x)
        oq, ot, oz = V.vq_encode(sd, cfg, x)
    z_ref = h.flatten(2).transpose(1, 2)
    assert torch.equal(ot, tokens), f"{name}: oracle tokens differ from upstream ({(ot != tokens).sum().item()} of {tokens.numel()})"
    assert (oz - z_ref).abs().max() < 1e-4 * z_ref.abs().max(), "latents"
    assert torch.allclose(oq, quant, atol=1e-6)
    print(f"[{name}] upstream==oracle: {tokens.numel()} tokens identical, latent max err {(oz - z_ref).abs().max():.2e}, loss {float(loss):.1f}")
    if check:
        return
    keys = list(ref.state_dict().keys())
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **{
        "meta/keys": np.array(keys), "meta/shapes": np.array([",".join(map(str, ref.state_dict()[k].shape)) for k in keys]),
        "meta/weight_checksum": np.array(sum(float(v.double().abs().sum()) for v in sd.values())),
        "meta/input_checksum": np.array(float(x.double().abs().sum())),
        "tokens": tokens.numpy().astype(np.int32), "latents": z_ref.numpy().astype(np.float32),
        "quant_sum": np.array(float(quant.double().sum()))})
    print("    wrote", f"{name}.npz")


def run_vqvae(name, check):
    """Training step of upstream's VQVAE (train mode: straight-through + commitment loss; EMA update on, dead-code replacement off) against
    torch autograd through the oracle: reconstruction, code loss and every parameter gradient."""
    c = VQVAE_CASES[name]
    nl, nlat = c.get("n_labels"), c.get("norm_latents", False)
    cfg = V.vq_cfg(c["enc_type"], image=c["image"], patch=c["patch"], codebook=c["codebook"], post_mlp=c["post_mlp"], channels=c.get("channels", 3),
                   patch_proj=c.get("patch_proj", True))
    sd = V.seeded_vqvae_state_dict(cfg, c["dec_type"], seed=c["seed"], n_labels=nl)
    x = V.synthetic_images(cfg, c["batch"], seed=c["seed"]) if nl is None else V.synthetic_labels(cfg, c["batch"], nl, seed=c["seed"])
    ref = RefVQVAE(dec_type=c["dec_type"], image_size=cfg.image, enc_type=c["enc_type"], patch_size=cfg.patch, post_mlp=cfg.post_mlp,
                   codebook_size=cfg.codebook, latent_dim=cfg.latent, norm_codes=True, sync_codebook=False, threshold_ema_dead_code=0,
                   commitment_weight=c["commitment_weight"], n_labels=nl, n_channels=cfg.channels, norm_latents=nlat, patch_proj=cfg.patch_proj)
    rec_loss = (lambda d: torch.nn.functional.mse_loss(d, x)) if nl is None else (lambda d: torch.nn.functional.cross_entropy(d, x))
    msg = ref.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    ref.train()
    dec, code_loss = ref(x)
    loss = rec_loss(dec) + code_loss.sum()
    loss.backward()
    grads = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    P = {k: v.clone().requires_grad_(k in grads) for k, v in sd.items()}
    odec, ocl, otok = V.vqvae_forward(P, cfg, c["dec_type"], x, commitment_weight=c["commitment_weight"], norm_latents=nlat)
    (rec_loss(odec) + ocl.sum()).backward()
    assert torch.allclose(odec, dec, atol=2e-5 * float(dec.abs().max())), float((odec - dec).abs().max())
    assert torch.allclose(ocl, code_loss, rtol=1e-5)
    worst = 0.0
    for k, g in grads.items():
        assert P[k].grad is not None, k
        err = float((P[k].grad - g).norm() / (g.norm() + 1e-20))
        worst = max(worst, err)
        assert err < 1e-4, (k, err)
    with torch.no_grad():
        ref.load_state_dict(sd, strict=True)          # (the training forward moved the codebook: EMA update)
        ref.eval()
        dt = ref.decode_tokens(otok)
        assert torch.allclose(V.vqvae_decode(sd, cfg, c["dec_type"], sd["quantize._codebook.embed"][otok].permute(0, 3, 1, 2)), dt, atol=2e-5 * float(dt.abs().max()))
    print(f"[{name}] upstream==oracle: dec max err {float((odec - dec).abs().max()):.2e}, code_loss {float(code_loss):.5f}, {len(grads)} gradients, worst rel err {worst:.1e}")
    if check:
        return
    keys = list(ref.state_dict().keys())
    fx = {"meta/keys": np.array(keys), "meta/shapes": np.array([",".join(map(str, ref.state_dict()[k].shape)) for k in keys]),
          "meta/grad_keys": np.array(list(grads)),
          "meta/weight_checksum": np.array(sum(float(v.double().abs().sum()) for v in sd.values())),
          "meta/input_checksum": np.array(float(x.double().abs().sum())),
          "tokens": otok.numpy().astype(np.int32), "code_loss": code_loss.detach().numpy(), "loss": np.array(float(loss)),
          "dec_fro": np.array(float(dec.double().norm())), "dec_tokens_fro": np.array(float(dt.double().norm()))}
    small = dec.numel() <= 1 << 16
    fx["dec" if small else "dec_head"] = (dec if small else dec[:, :, :8, :8]).detach().numpy().astype(np.float32)
    fx["dec_tokens" if small else "dec_tokens_head"] = (dt if small else dt[:, :, :8, :8]).numpy().astype(np.float32)
    for k, g in grads.items():
        fx["grad_l2/" + k] = np.array(float(g.double().norm()))
        fx["grad_head/" + k] = g.reshape(-1)[:16].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **fx)
    print("    wrote", f"{name}.npz")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("cases", nargs="*", default=list(VQ_CASES) + list(VQVAE_CASES))
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    for n in a.cases:
        (run_vqvae if n in VQVAE_CASES else run)(n, a.check)
