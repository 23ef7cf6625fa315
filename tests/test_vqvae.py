"""VQ-VAE: the ViT decoder behind the codebook and the gradient path of tokenizer training.
CPU: the oracle (torch autograd through oracle/vq_oracle.py) against the fixtures of the unmodified upstream VQVAE, state_dict layout.
GPU: decode_tokens, the training forward (dec, code_loss) and every parameter gradient of the HIP path against the same fixtures."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vq_oracle as V
from tests.golden.cases import VQVAE_CASES

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def case(name):
    c = VQVAE_CASES[name]
    nl = c.get("n_labels")
    cfg = V.vq_cfg(c["enc_type"], image=c["image"], patch=c["patch"], codebook=c["codebook"], post_mlp=c["post_mlp"], channels=c.get("channels", 3),
                   patch_proj=c.get("patch_proj", True))
    sd = V.seeded_vqvae_state_dict(cfg, c["dec_type"], seed=c["seed"], n_labels=nl)
    x = V.synthetic_images(cfg, c["batch"], seed=c["seed"]) if nl is None else V.synthetic_labels(cfg, c["batch"], nl, seed=c["seed"])
    return c, cfg, sd, x, np.load(os.path.join(GOLD, f"{name}.npz"))


def build(c, cfg, **kw):
    from fourm.vq import VQVAE
    return VQVAE(dec_type=c["dec_type"], image_size=cfg.image, enc_type=c["enc_type"], patch_size=cfg.patch, post_mlp=cfg.post_mlp,
                 codebook_size=cfg.codebook, latent_dim=cfg.latent, norm_codes=True, sync_codebook=False, threshold_ema_dead_code=0,
                 commitment_weight=c["commitment_weight"], n_labels=c.get("n_labels"), n_channels=cfg.channels, norm_latents=c.get("norm_latents", False), patch_proj=cfg.patch_proj, **kw)


def rec_loss(c, dec, x):
    """Reconstruction term of the trainers: MSE on pixels, cross-entropy over the class scores for segmentation maps."""
    return F.mse_loss(dec, x) if c.get("n_labels") is None else F.cross_entropy(dec, x)


@pytest.mark.parametrize("name", ["vqvae_small", "vqvae_semseg", "vqvae_feat"])
def test_oracle_training_step_matches_upstream_fixture(name):
    c, cfg, sd, x, g = case(name)
    assert sum(float(v.double().abs().sum()) for v in sd.values()) == pytest.approx(float(g["meta/weight_checksum"]), rel=1e-9)
    assert float(x.double().abs().sum()) == pytest.approx(float(g["meta/input_checksum"]), rel=1e-9)
    gk = set(g["meta/grad_keys"].tolist())
    P = {k: v.clone().requires_grad_(k in gk) for k, v in sd.items()}
    dec, cl, tok = V.vqvae_forward(P, cfg, c["dec_type"], x, commitment_weight=c["commitment_weight"], norm_latents=c.get("norm_latents", False))
    assert np.array_equal(tok.numpy(), g["tokens"])
    full = "dec" in g.files                                    # (large reconstructions are kept as their 8 x 8 corner + Frobenius norm)
    ref_dec = g["dec"] if full else g["dec_head"]
    np.testing.assert_allclose((dec if full else dec[:, :, :8, :8]).detach().numpy(), ref_dec, atol=2e-5 * float(np.abs(ref_dec).max()), rtol=0)
    assert float(dec.detach().double().norm()) == pytest.approx(float(g["dec_fro"]), rel=1e-5)
    np.testing.assert_allclose(cl.detach().numpy(), g["code_loss"], rtol=1e-5)
    (rec_loss(c, dec, x) + cl.sum()).backward()
    for k in gk:
        assert float(P[k].grad.double().norm()) == pytest.approx(float(g["grad_l2/" + k]), rel=1e-4, abs=1e-9), k
        np.testing.assert_allclose(P[k].grad.reshape(-1)[:16].numpy(), g["grad_head/" + k], rtol=1e-3, atol=1e-7)
    with torch.no_grad():
        dt = V.vqvae_decode(sd, cfg, c["dec_type"], sd["quantize._codebook.embed"][tok].permute(0, 3, 1, 2))
    ref_dt = g["dec_tokens"] if full else g["dec_tokens_head"]
    np.testing.assert_allclose((dt if full else dt[:, :, :8, :8]).numpy(), ref_dt, atol=2e-5 * float(np.abs(ref_dt).max()), rtol=0)


@pytest.mark.parametrize("name", list(VQVAE_CASES))
def test_vqvae_state_dict_layout(name):
    c, cfg, sd, x, g = case(name)
    m = build(c, cfg)
    msd = m.state_dict()
    shapes = dict(zip(g["meta/keys"].tolist(), g["meta/shapes"].tolist()))
    assert set(msd) == set(shapes)
    for k, v in msd.items():
        assert ",".join(map(str, v.shape)) == shapes[k], k
    assert not m.load_state_dict(sd, strict=True).missing_keys
    with pytest.raises(RuntimeError):
        m(x)                                     # no CPU fallback: the tokenizer computes through libfourm_hip.so on an MI355X


def _rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VQVAE_CASES))
def test_decode_tokens_matches_upstream_fixture(name):
    c, cfg, sd, x, g = case(name)
    m = build(c, cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    tok = torch.from_numpy(g["tokens"]).long().cuda()
    dt = m.decode_tokens(tok).cpu()
    assert dt.shape == ((x.shape[0], c["n_labels"], *x.shape[1:]) if c.get("n_labels") else x.shape) and dt.dtype == torch.float32
    # bf16 GEMM operands in the 8 / 12 decoder blocks (upstream's autocast arithmetic) against upstream's fp32 run
    assert float(dt.double().norm()) == pytest.approx(float(g["dec_tokens_fro"]), rel=5e-3)
    ref = torch.from_numpy(g["dec_tokens"] if "dec_tokens" in g else g["dec_tokens_head"])
    got = dt if "dec_tokens" in g else dt[:, :, :8, :8]
    assert _rel(got, ref) < 1.5e-2, _rel(got, ref)
    q = m.tokens_to_embedding(tok)
    assert torch.equal(m.decode_quant(q).cpu(), dt)                       # the two entry points share one path


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(VQVAE_CASES))
def test_training_step_matches_upstream_fixture(name):
    c, cfg, sd, x, g = case(name)
    m = build(c, cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    xg = x.cuda()
    dec, cl = m(xg)
    assert dec.requires_grad and cl.requires_grad and dec.shape[0] == x.shape[0] and dec.shape[-2:] == x.shape[-2:]
    loss = rec_loss(c, dec, xg) + cl.sum()
    loss.backward()
    # forward values (fp32 upstream; ours: bf16 operands in the blocks)
    ref_dec = torch.from_numpy(g["dec"] if "dec" in g else g["dec_head"])
    got_dec = dec.detach().cpu() if "dec" in g else dec.detach().cpu()[:, :, :8, :8]
    e_dec = _rel(got_dec, ref_dec)
    e_cl = abs(float(cl.detach()) - float(g["code_loss"][0])) / float(g["code_loss"][0])
    assert float(loss.detach()) == pytest.approx(float(g["loss"]), rel=1e-2)
    errs = {}
    for k, p in m.named_parameters():
        if "grad_l2/" + k not in g.files:
            assert p.grad is None or not p.requires_grad, k
            continue
        assert p.grad is not None and p.grad.dtype == torch.float32, k
        ref = float(g["grad_l2/" + k])
        errs[k] = abs(float(p.grad.double().norm()) - ref) / (ref + 1e-12)
    worst = max(errs, key=errs.get)
    med = float(np.median(list(errs.values())))
    from tests.parity_log import record as log
    log("vqvae_train_step", case=name, dec_rel=e_dec, code_loss_rel=e_cl, grad_norm_rel_worst=errs[worst], grad_norm_rel_median=med, worst=worst)
    assert e_dec < 2e-2 and e_cl < 1e-2, (e_dec, e_cl)
    assert errs[worst] < 8e-2 and med < 2e-2, (worst, errs[worst], med)
    if cfg.dim <= 512:
        # small models: every gradient TENSOR against torch autograd through the oracle (fp32, pinned to upstream by the fixture test above)
        gk = set(g["meta/grad_keys"].tolist())
        P = {k: v.clone().requires_grad_(k in gk) for k, v in sd.items()}
        odec, ocl, _ = V.vqvae_forward(P, cfg, c["dec_type"], x, commitment_weight=c["commitment_weight"], norm_latents=c.get("norm_latents", False))
        (rec_loss(c, odec, x) + ocl.sum()).backward()
        full = {k: _rel(p.grad.cpu(), P[k].grad) for k, p in m.named_parameters() if k in gk and float(P[k].grad.norm()) > 1e-10}
        wk = max(full, key=full.get)
        log("vqvae_train_step.full_tensors", case=name, grad_rel_worst=full[wk], worst=wk, grad_rel_median=float(np.median(list(full.values()))))
        assert full[wk] < 6e-2 and float(np.median(list(full.values()))) < 2.5e-2, (wk, full[wk])
    else:
        # gradient heads of the largest tensors on both sides of the quantizer (direction, not only norm)
        for k in ("decoder.out_proj.weight", "post_quant_proj.weight", "quant_proj.weight", "encoder.proj.weight", "encoder.blocks.0.attn.qkv.weight"):
            got = dict(m.named_parameters())[k].grad.reshape(-1)[:16].cpu()
            ref = torch.from_numpy(g["grad_head/" + k])
            assert _rel(got, ref) < 0.15, (k, _rel(got, ref))
    # the EMA codebook update ran once, after the code assignment
    assert float(m.quantize._codebook.cluster_size.sum()) == pytest.approx((1 - m.quantize._codebook.decay) * x.shape[0] * cfg.grid ** 2, rel=1e-5)


@pytest.mark.gpu
def test_gradients_accumulate_and_frozen_encoder():
    c, cfg, sd, x, g = case("vqvae_small")
    m = build(c, cfg)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    m.quantize.eval()                                                    # keep the codebook fixed: two identical steps
    xg = x.cuda()
    (F.mse_loss(m(xg)[0], xg)).backward()
    g1 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    assert "encoder.proj.weight" in g1 and "decoder.out_proj.bias" in g1
    (F.mse_loss(m(xg)[0], xg)).backward()                                # no zero_grad: torch semantics, gradients add up
    for k, p in m.named_parameters():
        if k in g1:
            assert torch.allclose(p.grad, 2 * g1[k], rtol=1e-3, atol=1e-6 * float(g1[k].abs().max())), k
    m.zero_grad(set_to_none=True)
    (F.mse_loss(m(xg)[0], xg)).backward()
    for k, p in m.named_parameters():
        if k in g1:
            assert torch.allclose(p.grad, g1[k], rtol=1e-3, atol=1e-6 * float(g1[k].abs().max())), k
    # freeze_enc: only the decoder side learns (vqvae.py:477-478)
    f = build(c, cfg, freeze_enc=True)
    f.load_state_dict(sd, strict=True)
    f = f.cuda().train()
    dec, cl = f(xg)
    F.mse_loss(dec, xg).backward()
    assert f.encoder.proj.weight.grad is None and f.quant_proj.weight.grad is None
    assert torch.allclose(f.decoder.out_proj.weight.grad, g1["decoder.out_proj.weight"], rtol=1e-3, atol=1e-6)
    with torch.no_grad():
        assert not f(xg)[0].requires_grad
