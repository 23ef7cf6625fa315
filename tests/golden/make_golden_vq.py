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
from fourm.vq.vqvae import VQ as RefVQ  # noqa: E402

from oracle import vq_oracle as V  # noqa: E402
from tests.golden.cases import VQ_CASES  # noqa: E402


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


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("cases", nargs="*", default=list(VQ_CASES))
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    for n in a.cases:
        run(n, a.check)
