"""Euclidean codebook (``VQ(norm_codes=False)``; upstream EuclideanCodebook, fourm/vq/quantizers/quantize_lucid.py:181-301).
CPU: the oracle restatement against the fixture dumped from the unmodified upstream class (tests/golden/make_golden_vq_euclid.py).
GPU: the HIP codebook (fm_vq_assign_bias / fm_vq_code_stats_raw / fm_vq_ema_update_euclid) against fixture and oracle."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
from oracle import vq_oracle as V  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def latents(R=1568, D=32):          # = tests/golden/make_golden_vq_euclid.py latents()
    g = torch.Generator().manual_seed(11)
    return torch.randn(R, D, generator=g) * 0.4, torch.randn(R, D, generator=g) * 0.4 + 0.05


def fixture():
    return np.load(os.path.join(GOLD, "vq_euclid.npz"))


def test_oracle_reproduces_upstream_fixture():
    fx = fixture()
    z1, z2 = latents(int(fx["R"]), int(fx["D"]))
    embed0 = torch.from_numpy(fx["embed0"])
    ind, q, dist = V.assign_codes_euclid(z1, embed0)
    assert np.array_equal(ind.numpy(), fx["ind1"])
    assert torch.equal(q, embed0[ind])
    emb, avg, cluster = embed0.clone(), embed0.clone(), torch.zeros(int(fx["K"]))
    for step, z in enumerate((z1, z2)):
        ind, _, _ = V.assign_codes_euclid(z, emb)
        assert np.array_equal(ind.numpy(), fx[f"ind_train{step}"])
        emb, avg, cluster = V.codebook_ema_update_euclid(avg, cluster, z, ind, float(fx["decay"]), float(fx["eps"]))
        for name, got in (("embed", emb), ("embed_avg", avg), ("cluster", cluster)):
            want = torch.from_numpy(fx[f"{name}_after{step}"])
            assert float((got - want).abs().max()) < 2e-6 * float(want.abs().max()), (step, name)


def test_oracle_norm_latents_and_kmeans_reproduce_upstream_fixture():
    """norm_latents=True (the codebook sees l2norm(z): EMA sums over the NORMALISED latents) and kmeans_init=True, both dumped from the
    unmodified upstream classes."""
    fx = fixture()
    K, D = int(fx["K"]), int(fx["D"])
    z1, z2 = latents(int(fx["R"]), D)
    embed0 = torch.from_numpy(fx["embed0"])
    emb, avg, cluster = embed0.clone(), embed0.clone(), torch.zeros(K)
    for step, z in enumerate((z1, z2)):
        zn = torch.nn.functional.normalize(z[: 4 * 196], p=2, dim=-1)
        ind, _, _ = V.assign_codes_euclid(zn, emb)
        assert np.array_equal(ind.numpy(), fx[f"nl_ind{step}"])
        emb, avg, cluster = V.codebook_ema_update_euclid(avg, cluster, zn, ind, float(fx["decay"]), float(fx["eps"]))
        for name, got in (("embed", emb), ("embed_avg", avg), ("cluster", cluster)):
            want = torch.from_numpy(fx[f"nl_{name}_after{step}"])
            assert float((got - want).abs().max()) < 2e-6 * float(want.abs().max()), (step, name)
    means, bins = V.kmeans_euclid(z1, torch.from_numpy(fx["km_init_index"]), int(fx["km_iters"]))
    assert float((means - torch.from_numpy(fx["km_embed"])).abs().max()) < 1e-6 and np.array_equal(bins.float().numpy(), fx["km_cluster"])


def test_constructor_builds_the_euclidean_codebook_with_upstream_buffers():
    from fourm.vq import VQ
    from fourm.vq.quantizers.quantize_lucid import EuclideanCodebook
    m = VQ(image_size=32, enc_type="vit_s_enc", patch_size=8, codebook_size=64, latent_dim=32, norm_codes=False, post_mlp=True)
    cb = m.quantize._codebook
    assert isinstance(cb, EuclideanCodebook)
    keys = {k for k in m.state_dict() if k.startswith("quantize.")}
    assert keys == {"quantize._codebook.initted", "quantize._codebook.cluster_size", "quantize._codebook.embed_avg", "quantize._codebook.embed"}
    assert torch.equal(cb.embed, cb.embed_avg) and bool(cb.initted)
    # kmeans_init=True constructs like upstream (zeros, initted = 0: quantize_lucid.py:196-208), so a checkpoint of such a tokenizer loads
    ck = EuclideanCodebook(dim=32, codebook_size=64, kmeans_init=True)
    assert not bool(ck.initted) and float(ck.embed.abs().max()) == 0 and float(ck.embed_avg.abs().max()) == 0
    ck.load_state_dict(cb.state_dict())
    assert bool(ck.initted) and torch.equal(ck.embed, cb.embed)


@pytest.mark.gpu
def test_hip_assignment_and_ema_match_upstream():
    from fourm.hip import _lib as L, ops
    from fourm.vq.quantizers.quantize_lucid import EuclideanCodebook
    fx = fixture()
    K, D, R = int(fx["K"]), int(fx["D"]), int(fx["R"])
    z1, z2 = (t.cuda() for t in latents(R, D))
    cb = EuclideanCodebook(dim=D, codebook_size=K, decay=float(fx["decay"]), eps=float(fx["eps"]), threshold_ema_dead_code=0).cuda()
    cb.embed.copy_(torch.from_numpy(fx["embed0"])); cb.embed_avg.copy_(cb.embed)

    def assign(z):
        splits = 1
        wv, wi = torch.empty(R, splits, device="cuda"), torch.empty(R, splits, dtype=torch.int32, device="cuda")
        tok = torch.empty(R, dtype=torch.int64, device="cuda")
        L.check(L.vq_assign_bias(ops._p(z), z.stride(0), ops._p(cb.embed), ops._p(cb.code_bias()), ops._p(cb.embed), K, D, R, 1, 0,
                                 ops._p(wv), ops._p(wi), splits, ops._p(tok), None, ops._stream()))
        return tok
    tok = assign(z1).cpu().numpy()
    diff = tok != fx["ind1"]
    # the same arg-max up to the rounding of two different fp32 formulas: a differing row must be a near tie of upstream's own scores
    assert diff.mean() < 2e-3 and (not diff.any() or float(fx["margin1"][diff].max()) < 1e-5), (diff.sum(), fx["margin1"][diff] if diff.any() else None)
    for step, z in enumerate((z1, z2)):
        ind = torch.from_numpy(fx[f"ind_train{step}"]).cuda()          # upstream's own indices: isolates the EMA arithmetic
        cb.ema_update_(z, ind)
        for name, got in (("embed", cb.embed), ("embed_avg", cb.embed_avg), ("cluster", cb.cluster_size)):
            want = torch.from_numpy(fx[f"{name}_after{step}"]).cuda()
            assert float((got - want).abs().max()) < 5e-6 * float(want.abs().max()), (step, name)


@pytest.mark.gpu
def test_hip_ema_with_normalised_latents_and_kmeans_init_match_upstream():
    """VectorQuantize(norm_latents=True) over the Euclidean codebook accumulates l2norm(z); kmeans_init=True runs upstream's
    Euclidean k-means on the first batch.  Both against the fixture of the unmodified classes."""
    from fourm.vq.quantizers.quantize_lucid import EuclideanCodebook
    fx = fixture()
    K, D, R = int(fx["K"]), int(fx["D"]), int(fx["R"])
    z1, z2 = (t.cuda() for t in latents(R, D))
    cb = EuclideanCodebook(dim=D, codebook_size=K, decay=float(fx["decay"]), eps=float(fx["eps"]), threshold_ema_dead_code=0).cuda()
    cb.embed.copy_(torch.from_numpy(fx["embed0"])); cb.embed_avg.copy_(cb.embed)
    for step, z in enumerate((z1, z2)):
        ind = torch.from_numpy(fx[f"nl_ind{step}"]).cuda()
        cb.ema_update_(z[: 4 * 196].contiguous(), ind, normalize=True)
        for name, got in (("embed", cb.embed), ("embed_avg", cb.embed_avg), ("cluster", cb.cluster_size)):
            want = torch.from_numpy(fx[f"nl_{name}_after{step}"]).cuda()
            assert float((got - want).abs().max()) < 5e-6 * float(want.abs().max()), (step, name)
    ck = EuclideanCodebook(dim=D, codebook_size=int(fx["km_K"]), kmeans_init=True, kmeans_iters=int(fx["km_iters"]), threshold_ema_dead_code=0).cuda()
    ck.init_embed_(z1, init_index=torch.from_numpy(fx["km_init_index"]).cuda())
    assert bool(ck.initted)
    want = torch.from_numpy(fx["km_embed"]).cuda()
    # (a near-tie of two means may move one sample between clusters under the kernel's <z, e> - |e|^2 / 2 form: means to 1e-3, counts to 2)
    assert float((ck.embed - want).abs().max()) < 1e-3 * float(want.abs().max()), float((ck.embed - want).abs().max())
    assert float((ck.cluster_size - torch.from_numpy(fx["km_cluster"]).cuda()).abs().max()) <= 2
    assert torch.equal(ck.embed, ck.embed_avg)


@pytest.mark.gpu
def test_vq_tokenizer_with_norm_codes_false():
    """VQ(norm_codes=False) end to end: the tokens are the Euclidean nearest codes of the latents the encoder produced, the quantised map
    is the codebook rows, and a VQ-VAE training forward moves the codebook by the EMA rule and reports upstream's commitment loss."""
    from fourm.vq import VQ, VQVAE
    torch.manual_seed(0)
    m = VQ(image_size=64, enc_type="vit_s_enc", patch_size=8, codebook_size=256, latent_dim=32, norm_codes=False, post_mlp=True).cuda().eval()
    x = torch.randn(4, 3, 64, 64, device="cuda")
    quant, _, tokens = m.encode(x)
    z = m._last_latents.reshape(-1, 32).cpu()
    ind, q, dist = V.assign_codes_euclid(z, m.quantize.codebook.cpu())
    top2 = dist.topk(2, dim=-1).values
    diff = tokens.reshape(-1).cpu() != ind
    assert diff.float().mean() < 0.01 and (not bool(diff.any()) or float((top2[:, 0] - top2[:, 1])[diff].max()) < 1e-4)
    assert torch.equal(quant.permute(0, 2, 3, 1).reshape(-1, 32).cpu(), m.quantize.codebook.cpu()[tokens.reshape(-1).cpu()])
    vae = VQVAE(image_size=64, enc_type="vit_s_enc", dec_type="vit_s_dec", patch_size=8, codebook_size=256, latent_dim=32, norm_codes=False, post_mlp=True,
                ema_decay=0.9, threshold_ema_dead_code=0).cuda().train()
    cb = vae.quantize._codebook
    e0, a0, c0 = cb.embed.clone().cpu(), cb.embed_avg.clone().cpu(), cb.cluster_size.clone().cpu()
    dec, code_loss = vae(x)
    (dec.float().pow(2).mean() + code_loss.sum()).backward()
    z = vae._last_latents.reshape(-1, 32).cpu() if getattr(vae, "_last_latents", None) is not None else None
    assert dec.shape == x.shape and float(code_loss) > 0
    assert not torch.equal(cb.embed.cpu(), e0) and float(cb.cluster_size.sum()) > 0
    assert any(p.grad is not None and float(p.grad.abs().sum()) > 0 for p in vae.encoder.parameters())
