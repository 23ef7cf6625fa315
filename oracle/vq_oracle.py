"""CPU oracle for the VQ tokenizer front end (image -> codes).  TEST INFRASTRUCTURE ONLY (see
fourm_oracle.py for the rules).  Functional fp32 PyTorch restatement of

    VQ.encode ................ fourm/vq/vqvae.py:302-318
    ViTEncoder.forward ....... fourm/vq/models/vit_models.py:465-501 (+ Attention :177-197, Mlp :155-162, Block :243-246)
    VectorQuantize.forward ... fourm/vq/quantizers/quantize_lucid.py:504-568 (eval branch)
    CosineSimCodebook.forward  fourm/vq/quantizers/quantize_lucid.py:388-407

    VQVAE.forward ............ fourm/vq/vqvae.py:454-481 (training branch of the quantizer: quantize_lucid.py:533-541)
    ViTDecoder.forward ....... fourm/vq/models/vit_models.py:617-648

Parity status: PINNED by tests/golden/make_golden_vq.py against the unmodified upstream ``VQ`` / ``VQVAE`` (same seeded
weights and images): tokens identical, latents to 1e-5; VQVAE reconstruction, code loss and every parameter gradient (torch autograd
through this restatement vs autograd through upstream) to 1e-5."""
import math
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F

from .fourm_oracle import _Num, seeded_tensor, sincos_2d

Tensor = torch.Tensor


@dataclass
class VQCfg:
    image: int = 224
    patch: int = 16
    channels: int = 3
    dim: int = 768
    depth: int = 12
    heads: int = 12
    mlp_ratio: float = 4.0
    post_mlp: bool = True
    patch_proj: bool = True    # False: feature-map tokenizer, 1 x 1 projection of (B, C, grid, grid) inputs
    codebook: int = 16384
    latent: int = 32
    eps: float = 1e-6

    @property
    def grid(self):
        return self.image // self.patch


def vq_cfg(enc_type: str, **kw) -> VQCfg:
    dim, depth, heads = {"vit_s_enc": (512, 8, 8), "vit_b_enc": (768, 12, 12), "vit_l_enc": (1024, 24, 16)}[enc_type]
    return VQCfg(dim=dim, depth=depth, heads=heads, **kw)


def seeded_vq_state_dict(cfg: VQCfg, seed: int = 0) -> Dict[str, Tensor]:
    D, Hd, g = cfg.dim, int(cfg.dim * cfg.mlp_ratio), cfg.grid
    sd: Dict[str, Tensor] = {}

    def lin(pre, o, i):
        sd[pre + ".weight"] = seeded_tensor(pre + ".weight", (o, i), 1.0 / math.sqrt(i), seed)
        sd[pre + ".bias"] = seeded_tensor(pre + ".bias", (o,), 0.02, seed)

    def norm(pre):
        sd[pre + ".weight"] = 1.0 + seeded_tensor(pre + ".weight", (D,), 0.1, seed)
        sd[pre + ".bias"] = seeded_tensor(pre + ".bias", (D,), 0.05, seed)
    sd["encoder.pos_emb"] = sincos_2d(g, g, D).reshape(g, g, D).permute(2, 0, 1)[None].contiguous()
    pp = cfg.patch if cfg.patch_proj else 1
    f = cfg.channels * pp * pp
    sd["encoder.proj.weight"] = seeded_tensor("encoder.proj.weight", (D, cfg.channels, pp, pp), 1.0 / math.sqrt(f), seed)
    sd["encoder.proj.bias"] = seeded_tensor("encoder.proj.bias", (D,), 0.02, seed)
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}"
        norm(p + ".norm1"); norm(p + ".norm2")
        lin(p + ".attn.qkv", 3 * D, D); lin(p + ".attn.proj", D, D)
        lin(p + ".mlp.fc1", Hd, D); lin(p + ".mlp.fc2", D, Hd)
    if cfg.post_mlp:
        norm("encoder.norm_mlp")
        lin("encoder.post_mlp.fc1", Hd, D); lin("encoder.post_mlp.fc2", D, Hd)
    sd["quant_proj.weight"] = seeded_tensor("quant_proj.weight", (cfg.latent, D, 1, 1), 1.0 / math.sqrt(D), seed)
    sd["quant_proj.bias"] = seeded_tensor("quant_proj.bias", (cfg.latent,), 0.02, seed)
    sd["quantize._codebook.initted"] = torch.Tensor([True])
    sd["quantize._codebook.cluster_size"] = torch.zeros(cfg.codebook)
    sd["quantize._codebook.embed"] = F.normalize(seeded_tensor("quantize._codebook.embed", (cfg.codebook, cfg.latent), 1.0, seed), dim=-1)
    return sd


def synthetic_images(cfg: VQCfg, batch: int, seed: int = 0) -> Tensor:
    g = torch.Generator().manual_seed(seed)
    side = cfg.image if cfg.patch_proj else cfg.grid           # feature-map tokenizers take (B, C, grid, grid)
    return torch.rand(batch, cfg.channels, side, side, generator=g) * 2 - 1


def vit_stack(P, pre0, t, dim, depth, heads, eps, post_mlp, num, tail):
    """The transformer blocks (+ tanh post-MLP) shared by ViTEncoder and ViTDecoder (Block :243-246; :494-496 / :635-636)."""
    B = t.shape[0]
    hd = dim // heads
    for i in range(depth):
        pre = f"{pre0}.blocks.{i}"
        h = num.layer_norm(t, P[pre + ".norm1.weight"], P[pre + ".norm1.bias"], eps)
        qkv = num.linear(h, P[pre + ".attn.qkv.weight"], P[pre + ".attn.qkv.bias"])
        q, k, v = [a.reshape(B, -1, heads, hd).transpose(1, 2) for a in qkv.chunk(3, -1)]
        s = num.r(num.r(num.r(q) @ num.r(k).transpose(-1, -2)) * hd ** -0.5)
        o = num.r(num.r(torch.softmax(s, -1)) @ num.r(v)).transpose(1, 2).reshape(B, -1, dim)
        t = t + num.linear(o, P[pre + ".attn.proj.weight"], P[pre + ".attn.proj.bias"])
        h = num.layer_norm(t, P[pre + ".norm2.weight"], P[pre + ".norm2.bias"], eps)
        h = num.act(num.linear(h, P[pre + ".mlp.fc1.weight"], P[pre + ".mlp.fc1.bias"]), "gelu")
        t = t + num.linear(h, P[pre + ".mlp.fc2.weight"], P[pre + ".mlp.fc2.bias"])
    if post_mlp:
        h = tail.layer_norm(t, P[pre0 + ".norm_mlp.weight"], P[pre0 + ".norm_mlp.bias"], eps)
        h = tail.r(torch.tanh(tail.linear(h, P[pre0 + ".post_mlp.fc1.weight"], P[pre0 + ".post_mlp.fc1.bias"])))
        t = t + tail.linear(h, P[pre0 + ".post_mlp.fc2.weight"], P[pre0 + ".post_mlp.fc2.bias"])
    return t


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def prepare_input(P, x: Tensor, undo_std: bool = False) -> Tensor:
    """VQ.prepare_input (vq/vqvae.py:269-286): ``2 * denormalize(x) - 1`` (denormalize = x * std + mean, fourm/utils/misc.py:23-37) and, when the
    state dict holds ``cls_emb.weight`` and x is an integer class map (B, H, W), the class embedding 'b h w c -> b c h w'."""
    if undo_std:
        mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
        x = 2.0 * (x * std + mean) - 1.0
    if not x.is_floating_point():
        x = P["cls_emb.weight"][x].permute(0, 3, 1, 2)
    return x


def synthetic_labels(cfg: VQCfg, batch: int, n_labels: int, seed: int = 0) -> Tensor:
    """Class maps (B, H, W) int64 in blocks of 4 x 4 pixels (segmentation-like)."""
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(0, n_labels, (batch, cfg.image // 4, cfg.image // 4), generator=g)
    return coarse.repeat_interleave(4, 1).repeat_interleave(4, 2).contiguous()


def vq_encode(P: Dict[str, Tensor], cfg: VQCfg, x: Tensor, emulate_bf16: bool = False, undo_std: bool = False):
    """Returns (quant (B, L, h, w), tokens (B, h, w) int64, latents z (B, h*w, L) before normalisation).
    ``emulate_bf16`` rounds at upstream's autocast points: the patch projection and the 12 blocks.  The post-MLP (autocast
    disabled, vit_models.py:494-496), the 1x1 projection and the codebook search (quantize_lucid.py:388-390) stay fp32."""
    num, tail = _Num(emulate_bf16), _Num(False)
    x = prepare_input(P, x, undo_std)
    B, C, H, W = x.shape
    p = cfg.patch if cfg.patch_proj else 1
    g = H // p
    # Conv2d(k = s = p): patches ordered (c, py, px) against weight.view(D, -1)
    patches = x.reshape(B, C, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, C * p * p)
    t = num.linear(patches, P["encoder.proj.weight"].reshape(cfg.dim, -1), P["encoder.proj.bias"])
    t = t + P["encoder.pos_emb"][0].permute(1, 2, 0).reshape(g * g, cfg.dim)
    t = vit_stack(P, "encoder", t, cfg.dim, cfg.depth, cfg.heads, cfg.eps, cfg.post_mlp, num, tail)
    wq = P["quant_proj.weight"].reshape(cfg.latent, cfg.dim)
    z = tail.r(t) @ tail.r(wq).t() + P["quant_proj.bias"]
    tokens, quant = assign_codes(z, P["quantize._codebook.embed"])
    return quant.reshape(B, g, g, cfg.latent).permute(0, 3, 1, 2), tokens.reshape(B, g, g), z


def assign_codes(z: Tensor, embed: Tensor):
    """Nearest code by cosine similarity, first index on ties; quantised vector = the raw codebook row."""
    zn, en = F.normalize(z.float(), dim=-1), F.normalize(embed, dim=-1)
    ind = (zn @ en.t()).argmax(-1)
    return ind, embed[ind]


def codebook_ema_update(embed: Tensor, cluster_size: Tensor, z: Tensor, ind: Tensor, decay: float, threshold_dead: float = 0.0,
                        replace_rows: Tensor = None, world_bins: Tensor = None, world_sums: Tensor = None):
    """Training-mode branch of upstream ``CosineSimCodebook.forward`` (fourm/vq/quantizers/quantize_lucid.py:409-426) after the code
    assignment: returns (new embed (K, d), new cluster_size (K)).

        bins[k]      = #latents assigned to k                                   (:410, all-reduced over ranks :411)
        cluster_size = cluster_size * decay + bins * (1 - decay)                (:413, ema_inplace :56-57)
        embed_sum    = sum of the L2-NORMALISED latents per code                (:418, all-reduced :419)
        target[k]    = l2norm(embed_sum[k] / bins[k])  (codes that got latents)  |  l2norm(embed[k])  (codes that got none)   (:421-424)
        embed        = embed * decay + target * (1 - decay)                     (:425)
        expire_codes_ (:366-383, 'batch_random'): codes with cluster_size < threshold_dead take ``replace_rows`` = the L2-normalised
        latents upstream draws with sample_vectors (:62-70); the draw itself (torch.randperm) is the caller's.
    ``world_bins`` / ``world_sums``: contributions of the other ranks (sync_codebook), added before the update like the all-reduce."""
    K = embed.shape[0]
    zn = F.normalize(z.float().reshape(-1, z.shape[-1]), dim=-1)
    ind = ind.reshape(-1).long()
    bins = torch.bincount(ind, minlength=K).float()
    sums = torch.zeros(K, zn.shape[1]).index_add_(0, ind, zn)
    if world_bins is not None:
        bins = bins + world_bins
        sums = sums + world_sums
    cluster = cluster_size.clone().mul_(decay).add_(bins, alpha=1 - decay)
    zero = bins == 0
    target = F.normalize(sums / bins.masked_fill(zero, 1.0).unsqueeze(1), dim=-1)
    target = torch.where(zero[:, None], F.normalize(embed, dim=-1), target)
    new_embed = embed.clone().mul_(decay).add_(target, alpha=1 - decay)
    if threshold_dead > 0:
        dead = cluster < threshold_dead
        if bool(dead.any()):
            new_embed[dead] = replace_rows
    return new_embed, cluster


def assign_codes_euclid(z: Tensor, embed: Tensor):
    """Nearest code by Euclidean distance (upstream ``EuclideanCodebook.forward``, quantize_lucid.py:272-281): arg-max of
    -(|z|^2 - 2 z e^T + |e|^2), first index on ties; quantised vector = the codebook row.  -> (ind, quantize, dist (R, K))."""
    z = z.float().reshape(-1, z.shape[-1])
    et = embed.t()
    dist = -(z.pow(2).sum(1, keepdim=True) - 2 * z @ et + et.pow(2).sum(0, keepdim=True))
    ind = dist.argmax(-1)
    return ind, embed[ind], dist


def codebook_ema_update_euclid(embed_avg: Tensor, cluster_size: Tensor, z: Tensor, ind: Tensor, decay: float, eps: float = 1e-5,
                               threshold_dead: float = 0.0, replace_rows: Tensor = None):
    """Training-mode branch of upstream ``EuclideanCodebook.forward`` (quantize_lucid.py:282-297) after the code assignment:
        cluster_size = cluster_size * decay + bins * (1 - decay)                                  (:286, ema_inplace)
        embed_avg    = embed_avg * decay + (sum of the raw latents per code) * (1 - decay)        (:288-292)
        embed        = embed_avg / ((cluster_size + eps) / (sum(cluster_size) + K eps) * sum(cluster_size))    (:293-295, laplace_smoothing)
        expire_codes_ (:358-375): codes with cluster_size < threshold_dead take ``replace_rows`` (upstream L2-normalises its samples, :343-345).
    -> (new embed, new embed_avg, new cluster_size)."""
    K = embed_avg.shape[0]
    z = z.float().reshape(-1, z.shape[-1])
    ind = ind.reshape(-1).long()
    bins = torch.bincount(ind, minlength=K).float()
    sums = torch.zeros(K, z.shape[1]).index_add_(0, ind, z)
    cluster = cluster_size.clone().mul_(decay).add_(bins, alpha=1 - decay)
    avg = embed_avg.clone().mul_(decay).add_(sums, alpha=1 - decay)
    smoothed = (cluster + eps) / (cluster.sum() + K * eps) * cluster.sum()
    new_embed = avg / smoothed.unsqueeze(1)
    if threshold_dead > 0:
        dead = cluster < threshold_dead
        if bool(dead.any()):
            new_embed[dead] = replace_rows
    return new_embed, avg, cluster


# ---- VQ-VAE: decoder + training-mode quantizer ---------------------------------------------------------------------
DEC_DIMS = {"vit_s_dec": (512, 8, 8), "vit_b_dec": (768, 12, 12), "vit_l_dec": (1024, 24, 16)}


def seeded_vqvae_state_dict(cfg: VQCfg, dec_type: str, seed: int = 0, n_labels: int = None) -> Dict[str, Tensor]:
    """Encoder / quantizer as seeded_vq_state_dict + ``decoder.*`` and ``post_quant_proj.*`` in upstream's state_dict layout."""
    sd = seeded_vq_state_dict(cfg, seed)
    D, depth, _ = DEC_DIMS[dec_type]
    Hd, g = int(D * cfg.mlp_ratio), cfg.grid

    def lin(pre, o, i):
        sd[pre + ".weight"] = seeded_tensor(pre + ".weight", (o, i), 1.0 / math.sqrt(i), seed)
        sd[pre + ".bias"] = seeded_tensor(pre + ".bias", (o,), 0.02, seed)

    def norm(pre):
        sd[pre + ".weight"] = 1.0 + seeded_tensor(pre + ".weight", (D,), 0.1, seed)
        sd[pre + ".bias"] = seeded_tensor(pre + ".bias", (D,), 0.05, seed)
    sd["decoder.pos_emb"] = sincos_2d(g, g, D).reshape(g, g, D).permute(2, 0, 1)[None].contiguous()
    for i in range(depth):
        p = f"decoder.blocks.{i}"
        norm(p + ".norm1"); norm(p + ".norm2")
        lin(p + ".attn.qkv", 3 * D, D); lin(p + ".attn.proj", D, D)
        lin(p + ".mlp.fc1", Hd, D); lin(p + ".mlp.fc2", D, Hd)
    if cfg.post_mlp:
        norm("decoder.norm_mlp")
        lin("decoder.post_mlp.fc1", Hd, D); lin("decoder.post_mlp.fc2", D, Hd)
    lin("decoder.out_proj", (n_labels or cfg.channels) * (cfg.patch * cfg.patch if cfg.patch_proj else 1), D)
    if n_labels:
        sd["cls_emb.weight"] = seeded_tensor("cls_emb.weight", (n_labels, cfg.channels), 1.0, seed)
    sd["post_quant_proj.weight"] = seeded_tensor("post_quant_proj.weight", (D, cfg.latent, 1, 1), 1.0 / math.sqrt(cfg.latent), seed)
    sd["post_quant_proj.bias"] = seeded_tensor("post_quant_proj.bias", (D,), 0.02, seed)
    return sd


def vqvae_decode(P, cfg: VQCfg, dec_type: str, quant: Tensor, emulate_bf16: bool = False) -> Tensor:
    """quant (B, L, h, w) -> image (B, C, H, W)   [VQVAE.decode_quant :454-465, ViTDecoder.forward :617-648]"""
    num, tail = _Num(emulate_bf16), _Num(False)
    D, depth, heads = DEC_DIMS[dec_type]
    B, Ld, g, _ = quant.shape
    t = quant.permute(0, 2, 3, 1).reshape(B, g * g, Ld) @ P["post_quant_proj.weight"].reshape(D, Ld).t() + P["post_quant_proj.bias"]
    t = t + P["decoder.pos_emb"][0].permute(1, 2, 0).reshape(g * g, D)
    t = vit_stack(P, "decoder", t, D, depth, heads, cfg.eps, cfg.post_mlp, num, tail)
    rows = num.linear(t, P["decoder.out_proj.weight"], P["decoder.out_proj.bias"]).float()
    p = cfg.patch if cfg.patch_proj else 1
    C = P["decoder.out_proj.weight"].shape[0] // (p * p)          # n_channels, or n_labels for class maps
    return rows.reshape(B, g, g, C, p, p).permute(0, 3, 1, 4, 2, 5).reshape(B, C, g * p, g * p)


def vqvae_forward(P, cfg: VQCfg, dec_type: str, x: Tensor, commitment_weight: float = 1.0, emulate_bf16: bool = False, norm_latents: bool = False):
    """Training-mode ``VQVAE.forward``: (dec, code_loss (1,), tokens).  Differentiable w.r.t. P: the quantised code passes its gradient
    straight to the latents (quantize = z + (q - z).detach()) and code_loss = w * mse(q.detach(), z)   [quantize_lucid.py:533-541]."""
    _, tokens, z = vq_encode(P, cfg, x, emulate_bf16)
    B, g = x.shape[0], x.shape[2] // (cfg.patch if cfg.patch_proj else 1)
    q = P["quantize._codebook.embed"][tokens.reshape(B, -1)].detach()
    if norm_latents:                      # the latents are normalised in front of the codebook and the commitment term (:525-527)
        z = F.normalize(z, p=2, dim=-1)
    quant = z + (q - z).detach()
    code_loss = (F.mse_loss(q, z) * commitment_weight).reshape(1)
    dec = vqvae_decode(P, cfg, dec_type, quant.reshape(B, g, g, cfg.latent).permute(0, 3, 1, 2), emulate_bf16)
    return dec, code_loss, tokens


def kmeans_cosine(samples: Tensor, init_index: Tensor, num_iters: int = 10):
    """Upstream ``kmeans`` (quantize_lucid.py:137-167) with use_cosine_sim=True and the sampled initial means made explicit
    (``means = samples[init_index]`` for sample_fn).  samples: l2-normalised rows.  Returns (means (K, d), bins (K) int64)."""
    means = samples[init_index]
    K, d = means.shape
    bins = None
    for _ in range(num_iters):
        buckets = (samples @ means.t()).argmax(-1)
        bins = torch.bincount(buckets, minlength=K)
        zero = bins == 0
        new = torch.zeros(K, d, dtype=samples.dtype).scatter_add_(0, buckets[:, None].expand(-1, d), samples)
        new = F.normalize(new / bins.masked_fill(zero, 1)[:, None], p=2, dim=-1)
        means = torch.where(zero[:, None], means, new)
    return means, bins


def kmeans_euclid(samples: Tensor, init_index: Tensor, num_iters: int = 10):
    """Upstream ``kmeans`` (quantize_lucid.py:137-167) with use_cosine_sim=False (``EuclideanCodebook.init_embed_``, :220-231) and the
    sampled initial means made explicit: nearest mean by -(x - m)^2 (first index on ties), per-cluster mean, empty clusters keep theirs.
    Returns (means (K, d), bins (K) int64)."""
    means = samples[init_index]
    K, d = means.shape
    bins = None
    for _ in range(num_iters):
        dists = -((samples[:, None, :] - means[None, :, :]) ** 2).sum(-1)
        buckets = dists.argmax(-1)
        bins = torch.bincount(buckets, minlength=K)
        zero = bins == 0
        new = torch.zeros(K, d, dtype=samples.dtype).scatter_add_(0, buckets[:, None].expand(-1, d), samples)
        new = new / bins.masked_fill(zero, 1)[:, None]
        means = torch.where(zero[:, None], means, new)
    return means, bins
