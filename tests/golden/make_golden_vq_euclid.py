#!/usr/bin/env python3
"""Golden fixture for the Euclidean codebook from the UNMODIFIED upstream ``EuclideanCodebook`` / ``VectorQuantize(use_cosine_sim=False)``
(fourm/vq/quantizers/quantize_lucid.py:181-301, :432-568; container only): latents in, code indices, quantised rows, and the codebook
state after two training-mode forwards (EMA of cluster_size / embed_avg, Laplace-smoothed division).  The oracle restatement
(oracle/vq_oracle.py assign_codes_euclid / codebook_ema_update_euclid) must reproduce it before the file is written.
    python tests/golden/make_golden_vq_euclid.py [--check]"""
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
from fourm.vq.quantizers.quantize_lucid import EuclideanCodebook as RefEuclid, VectorQuantize as RefVQn  # noqa: E402

from oracle import vq_oracle as V  # noqa: E402


def latents(R=1568, D=32):
    """The two latent batches of the fixture (CPU generator: reproducible everywhere)."""
    g = torch.Generator().manual_seed(11)
    return torch.randn(R, D, generator=g) * 0.4, torch.randn(R, D, generator=g) * 0.4 + 0.05


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="compare with the committed fixture instead of writing it")
    a = ap.parse_args()
    torch.manual_seed(0)
    K, D, R, decay, eps = 512, 32, 1568, 0.9, 1e-5
    cb = RefEuclid(dim=D, codebook_size=K, decay=decay, eps=eps, threshold_ema_dead_code=0)
    embed0 = cb.embed.clone()
    z1, z2 = latents(R, D)
    fx = dict(K=K, D=D, R=R, decay=decay, eps=eps, embed0=embed0.numpy())      # (z1, z2: regenerated from the seed on the test side, see latents())
    # inference
    cb.eval()
    with torch.no_grad():
        q, ind = cb(z1)
    o_ind, o_q, dist = V.assign_codes_euclid(z1, embed0)
    assert torch.equal(ind, o_ind) and torch.equal(q, o_q), "oracle assignment differs from upstream"
    top2 = dist.topk(2, dim=-1).values
    fx.update(ind1=ind.numpy(), margin1=(top2[:, 0] - top2[:, 1]).numpy())
    # two training-mode forwards
    cb.train()
    avg, cluster, emb = cb.embed_avg.clone(), cb.cluster_size.clone(), cb.embed.clone()
    for step, z in enumerate((z1, z2)):
        with torch.no_grad():
            _, ind_t = cb(z)
        o_ind, _, _ = V.assign_codes_euclid(z, emb)
        assert torch.equal(ind_t, o_ind)
        emb, avg, cluster = V.codebook_ema_update_euclid(avg, cluster, z, o_ind, decay, eps)
        for name, a_, b_ in (("embed", cb.embed, emb), ("embed_avg", cb.embed_avg, avg), ("cluster_size", cb.cluster_size, cluster)):
            err = float((a_ - b_).abs().max() / (b_.abs().max() + 1e-30))
            assert err < 2e-6, (step, name, err)
        fx[f"ind_train{step}"] = ind_t.numpy()
        fx[f"embed_after{step}"], fx[f"embed_avg_after{step}"], fx[f"cluster_after{step}"] = cb.embed.numpy().copy(), cb.embed_avg.numpy().copy(), cb.cluster_size.numpy().copy()
    # the VectorQuantize wrapper: straight-through output and commitment loss of a training forward (image feature-map layout)
    vq = RefVQn(dim=D, codebook_size=K, decay=decay, eps=eps, use_cosine_sim=False, threshold_ema_dead_code=0, commitment_weight=1.0)
    vq._codebook.embed.copy_(embed0); vq._codebook.embed_avg.copy_(embed0)
    vq.train()
    x = z1[: 4 * 196].reshape(4, 14, 14, D).permute(0, 3, 1, 2).contiguous()
    quant, loss, tok = vq(x)
    oi, oq, _ = V.assign_codes_euclid(z1[: 4 * 196], embed0)
    assert torch.equal(tok.reshape(-1), oi)
    o_loss = torch.nn.functional.mse_loss(oq, z1[: 4 * 196])
    assert abs(float(loss) - float(o_loss)) < 1e-6 * float(o_loss)
    fx.update(vq_tokens=tok.numpy(), vq_loss=np.float32(float(loss)))
    # norm_latents=True with the Euclidean codebook (VectorQuantize.forward hands the codebook l2norm(x), :525-527): the EMA sums are over the
    # NORMALISED latents.  Two training forwards of the unmodified wrapper; the oracle follows with l2norm(z) in its Euclidean functions.
    vqn = RefVQn(dim=D, codebook_size=K, decay=decay, eps=eps, use_cosine_sim=False, threshold_ema_dead_code=0, commitment_weight=1.0, norm_latents=True)
    vqn._codebook.embed.copy_(embed0); vqn._codebook.embed_avg.copy_(embed0)
    vqn.train()
    avg, cluster, emb = embed0.clone(), torch.zeros(K), embed0.clone()
    for step, z in enumerate((z1, z2)):
        xs = z[: 4 * 196].reshape(4, 14, 14, D).permute(0, 3, 1, 2).contiguous()
        with torch.no_grad():
            _, loss_n, tok_n = vqn(xs)
        zn = torch.nn.functional.normalize(z[: 4 * 196], p=2, dim=-1)
        oi, oq, _ = V.assign_codes_euclid(zn, emb)
        assert torch.equal(tok_n.reshape(-1), oi), "oracle (norm_latents) assignment differs from upstream"
        assert abs(float(loss_n) - float(torch.nn.functional.mse_loss(oq, zn))) < 1e-6 * float(loss_n)
        emb, avg, cluster = V.codebook_ema_update_euclid(avg, cluster, zn, oi, decay, eps)
        for name, a_, b_ in (("embed", vqn._codebook.embed, emb), ("embed_avg", vqn._codebook.embed_avg, avg), ("cluster_size", vqn._codebook.cluster_size, cluster)):
            err = float((a_ - b_).abs().max() / (b_.abs().max() + 1e-30))
            assert err < 2e-6, ("norm_latents", step, name, err)
        fx[f"nl_ind{step}"] = tok_n.reshape(-1).numpy()
        fx[f"nl_embed_after{step}"], fx[f"nl_embed_avg_after{step}"], fx[f"nl_cluster_after{step}"] = (
            vqn._codebook.embed.numpy().copy(), vqn._codebook.embed_avg.numpy().copy(), vqn._codebook.cluster_size.numpy().copy())
    # kmeans_init=True: upstream's init_embed_ with the random sample of initial means replaced by a fixed index set
    Kk, iters = 64, 4
    gi = torch.Generator().manual_seed(5)
    init_index = torch.randperm(R, generator=gi)[:Kk]
    cbk = RefEuclid(dim=D, codebook_size=Kk, kmeans_init=True, kmeans_iters=iters, decay=decay, eps=eps, threshold_ema_dead_code=0)
    assert not bool(cbk.initted) and float(cbk.embed.abs().max()) == 0
    cbk.sample_fn = lambda samples, num: samples[init_index]
    cbk.init_embed_(z1)
    means, bins = V.kmeans_euclid(z1, init_index, iters)
    assert float((cbk.embed - means).abs().max()) < 1e-6 and torch.equal(cbk.cluster_size, bins.float()) and bool(cbk.initted)
    fx.update(km_K=Kk, km_iters=iters, km_init_index=init_index.numpy(), km_embed=cbk.embed.numpy().copy(), km_cluster=cbk.cluster_size.numpy().copy())
    path = os.path.join(HERE, "vq_euclid.npz")
    if a.check:
        old = np.load(path)
        for k, v in fx.items():
            assert np.array_equal(np.asarray(v), old[k]), k
        print("vq_euclid: fixture reproduced")
        return
    np.savez_compressed(path, **fx)
    print("wrote", path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
