"""Launch sequence of the tokenizer front end on one MI355X (inference): conv-as-GEMM patch projection,
the ViT blocks on the trunk's kernels (LayerNorm with bias, qkv GEMM + bias, unmasked fused attention,
bias+GELU MLP with fused residual adds), the tanh "post MLP", the 1x1 projection to the latent
dimension and the fused cosine-similarity code search.  [upstream vq/vqvae.py:302-318,
vq/models/vit_models.py:465-501, vq/quantizers/quantize_lucid.py:388-407, :504-568]"""
import weakref

import torch
import torch.nn.functional as F

from fourm.hip import _lib as L
from fourm.hip import ops
from fourm.hip.engine import FourMEngine, Workspace, ru


class _VitEngine(FourMEngine):
    """The trunk executor configured from a ViT encoder instead of a FourM model."""

    def __init__(self, enc):
        blk = enc.blocks[0]
        self.model = enc
        self.D, self.H = enc.dim_tokens, blk.attn.num_heads
        if self.D // self.H != 64:
            raise NotImplementedError("the HIP attention kernels are built for head_dim 64")
        self.gated, self.act, self.qk_norm = False, "gelu", False
        self.fp32, self.adt = False, torch.bfloat16       # the 12 blocks follow autocast; the post-MLP / projection call the fp32 GEMM
        self.Hd = blk.mlp.hidden_features
        self.Hp = ru(self.Hd, 64)
        self.scale, self.eps = 64 ** -0.5, blk.norm1.eps
        self.ws, self.shadows, self._shadow_table, self._ctx, self.reducer = None, {}, None, None, None
        self.flat_params = self.flat_grads = None
        self._cache, self._dw_jobs = {}, None

    @property
    def device(self):
        return self.model.proj.weight.device


def _engine(enc) -> _VitEngine:
    eng = getattr(enc, "_hip_engine", None)
    if eng is None or eng.device != enc.proj.weight.device:
        if not enc.proj.weight.is_cuda:
            raise RuntimeError("the tokenizer computes on an MI355X through libfourm_hip.so; move it to the GPU first")
        eng = _VitEngine(enc)
        object.__setattr__(enc, "_hip_engine", eng)
    eng.prepare()
    return eng


def _tokens(enc, x):
    """(B, C, H, W) -> fp32 residual stream (Rp, D) after the last block (+ post MLP), plus (B, n_h, n_w)."""
    eng = _engine(enc)
    ws, D = eng.ws, eng.D
    B, C, Hh, Ww = x.shape
    P = enc.P_H
    assert enc.P_H == enc.P_W and Hh % P == 0 and Ww % P == 0, f"Image sizes {Hh}x{Ww} must be divisible by patch size {P}"
    nh, nw = Hh // P, Ww // P
    G, R = nh * nw, B * nh * nw
    Rp = ru(R, 128)
    x = x.float().contiguous()
    feat = C * P * P
    patches = ws.get("vq.patches", (Rp, ru(feat, 64)), torch.bfloat16)
    L.check(L.vq_patchify(ops._p(x), ops._p(patches), patches.stride(0), B, C, Hh, Ww, P, ops._stream()))
    # position table tiled over the batch (residual operand of the projection GEMM)
    key = ("pos", B, nh, nw, enc.pos_emb._version, enc.pos_emb.data_ptr())
    pos = eng._cache.get(key)
    if pos is None:
        pe = enc.pos_emb
        if pe.shape[-2:] != (nh, nw):
            pe = F.interpolate(pe, size=(nh, nw), mode="bicubic", align_corners=False)
        pos = torch.zeros(Rp, D, dtype=torch.float32, device=x.device)
        pos[:R] = pe[0].permute(1, 2, 0).reshape(G, D).repeat(B, 1)
        eng._cache = {key: pos}
    stream = ws.get("vq.x0", (Rp, D), torch.float32)
    ops.gemm_nt(patches, eng.w(enc.proj.weight), stream, epilogue=L.EPI_RESIDUAL, res=pos, bias=enc.proj.bias, M=R, N=D, K=ru(feat, 64))
    none = dict(mask_kind=L.MASK_NONE)
    for i, blk in enumerate(enc.blocks):
        stream = eng.encoder_block_fwd(blk, stream, B, G, none, None, f"vit{i % 2}")
    if hasattr(enc, "post_mlp"):
        # x.float() + fc2(tanh(fc1(norm_mlp(x.float()))))  with autocast DISABLED upstream (vit_models.py:494-496): fp32 operands
        # on the fp32 matrix cores (fm_gemm_f32: v_mfma_f32_32x32x2_f32, exact fp32) - no bf16 rounding in this tail
        f32 = torch.float32
        n = ws.get("vq.n", (Rp, D), f32)
        ops.layernorm_fwd(stream, enc.norm_mlp.weight, enc.norm_mlp.bias, n, eps=enc.norm_mlp.eps, R=R)
        hid = enc.post_mlp.fc1.weight.shape[0]
        t = ws.get("vq.t", (Rp, hid), f32)
        ops.gemm_nt(n, enc.post_mlp.fc1.weight.detach(), t, epilogue=L.EPI_TANH, bias=enc.post_mlp.fc1.bias, M=R, N=hid, K=D)
        out = ws.get("vq.post", (Rp, D), f32)
        ops.gemm_nt(t, enc.post_mlp.fc2.weight.detach(), out, epilogue=L.EPI_RESIDUAL, res=stream, bias=enc.post_mlp.fc2.bias, M=R, N=D, K=hid)
        stream = out
    return eng, stream, (B, nh, nw)


@torch.no_grad()
def encoder_forward(enc, x):
    eng, stream, (B, nh, nw) = _tokens(enc, x)
    return stream[: B * nh * nw].view(B, nh, nw, eng.D).permute(0, 3, 1, 2).contiguous()


@torch.no_grad()
def vq_encode(vq, x):
    enc = vq.encoder
    eng, stream, (B, nh, nw) = _tokens(enc, x)
    ws, D, Ld = eng.ws, eng.D, vq.latent_dim
    G, R = nh * nw, B * nh * nw
    # 1x1 convolution to the latent dimension: fp32 like the codebook search that follows (the tokenization script runs without
    # autocast, save_vq_tokens.py; 0.2 % of the FLOPs)
    z = ws.get("vq.z", (stream.shape[0], Ld), torch.float32)
    ops.gemm_nt(stream, vq.quant_proj.weight.detach().reshape(Ld, D), z, epilogue=L.EPI_F32, bias=vq.quant_proj.bias, M=R, N=Ld, K=D)
    cb = vq.quantize._codebook
    K = cb.embed.shape[0]
    # ONE buffer of l2-normalised codes, recomputed in place when the codebook changed (in training mode every encode() moves the
    # codebook: a cache keyed on its version would keep every past copy alive)
    stamp = (cb.embed._version, cb.embed.data_ptr(), getattr(cb, "epoch", 0), tuple(cb.embed.shape))
    slot = eng._cache.get("codes")
    if slot is None or slot[0][1:] != stamp[1:] or slot[1].device != cb.embed.device:
        slot = eng._cache["codes"] = [None, torch.empty_like(cb.embed)]
    en = slot[1]
    if slot[0] != stamp:
        L.check(L.l2norm_rows(ops._p(cb.embed), cb.embed.stride(0), ops._p(en), en.stride(0), K, Ld, ops._stream()))
        slot[0] = stamp
    splits = max(1, min(16, K // 1024))
    wv = ws.get("vq.wv", (R, splits), torch.float32)
    wi = ws.get("vq.wi", (R, splits), torch.int32)
    tokens = torch.empty(B, nh, nw, dtype=torch.int64, device=x.device)
    quant = torch.empty(B, Ld, nh, nw, dtype=torch.float32, device=x.device)
    # cosine similarity normalises the latents (quantize_lucid.py:394-395); norm_latents only moves that
    # normalisation in front of the (training-time) commitment loss
    L.check(L.vq_assign(ops._p(z), z.stride(0), ops._p(en), ops._p(cb.embed), K, Ld, R, G, 1, ops._p(wv), ops._p(wi), splits,
                        ops._p(tokens), ops._p(quant), ops._stream()))
    vq._last_latents = z[:R].view(B, G, Ld)
    return quant, torch.zeros(1, device=x.device), tokens
