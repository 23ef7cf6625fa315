"""Launch sequence of the tokenizer front end on one MI355X (inference): conv-as-GEMM patch projection,
the ViT blocks on the trunk's kernels (LayerNorm with bias, qkv GEMM + bias, unmasked fused attention,
bias+GELU MLP with fused residual adds), the tanh "post MLP", the 1x1 projection to the latent
dimension and the fused cosine-similarity code search.  [upstream vq/vqvae.py:302-318,
vq/models/vit_models.py:465-501, vq/quantizers/quantize_lucid.py:388-407, :504-568]"""
import os
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
        self._grads = {}                     # id(param) -> fp32 accumulator (the tokenizer has no flat store: few, large tensors)

    @property
    def device(self):
        return self.model.blocks[0].norm1.weight.device

    def prepare(self):
        if self.ws is None or self.ws.device != self.device:
            self.ws = Workspace(self.device, per_stream=True)          # (sub-batches may be in flight on several streams: tokenize_sub_batches)

    # gradients of tokenizer training: one fp32 buffer per parameter, attached as ``param.grad`` after the backward
    def grad_view(self, p):
        g = self._grads.get(id(p))
        if g is None or g.device != p.device:
            g = self._grads[id(p)] = torch.zeros(p.shape, dtype=torch.float32, device=p.device)
        return g

    def open_window(self, params):
        """torch semantics: gradients accumulate until the trainer clears them.  A parameter whose ``.grad`` is no longer the buffer this
        engine attached (zero_grad(set_to_none=True), or never attached) starts from zero."""
        for p in params:
            if p.requires_grad:
                g = self.grad_view(p)
                if p.grad is not g:
                    g.zero_()

    def attach(self, params):
        for p in params:
            if p.requires_grad:
                p.grad = self.grad_view(p)


def _engine(enc) -> _VitEngine:
    eng = getattr(enc, "_hip_engine", None)
    dev = enc.blocks[0].norm1.weight.device
    if eng is None or eng.device != dev:
        if dev.type != "cuda":
            raise RuntimeError("the tokenizer computes on an MI355X through libfourm_hip.so; move it to the GPU first")
        eng = _VitEngine(enc)
        object.__setattr__(enc, "_hip_engine", eng)
    eng.prepare()
    return eng


def _pos_rows(eng, vit, B, nh, nw, Rp):
    """Position table tiled over the batch: (Rp, D) f32, the residual operand of the GEMM that opens the stack."""
    pe = vit.pos_emb
    key = ("pos", B, nh, nw, pe._version, pe.data_ptr())
    pos = eng._cache.get(key)
    if pos is None:
        t = pe.detach()
        if t.shape[-2:] != (nh, nw):
            t = F.interpolate(t, size=(nh, nw), mode="bicubic", align_corners=False)
        pos = torch.zeros(Rp, eng.D, dtype=torch.float32, device=t.device)
        pos[:B * nh * nw] = t[0].permute(1, 2, 0).reshape(nh * nw, eng.D).repeat(B, 1)
        # drop the tables of an OLDER position embedding only: one table per batch size stays (sub-batches of different sizes may be in flight on
        # several streams, fourm.vq.tokenize_sub_batches - freeing the other size's table under a running GEMM corrupted its residual operand)
        eng._cache = {k: v for k, v in eng._cache.items() if not (isinstance(k, tuple) and k[0] == "pos" and k[4:] != key[4:])}
        eng._cache[key] = pos
        torch.cuda.current_stream(pos.device).synchronize()        # (built once per batch size: complete before any other stream may pick it up)
    return pos


def _blocks_fwd(eng, vit, stream, B, G, st, prefix):
    none = dict(mask_kind=L.MASK_NONE)
    for i, blk in enumerate(vit.blocks):
        sv = {} if st is not None else None
        # (the last residual sum of every block but the final one is left to the next block's norm1: a bf16 GEMM + fm_layernorm_fwd_res)
        stream = eng.encoder_block_fwd(blk, stream, B, G, none, sv, f"{prefix}{i}" if st is not None else f"{prefix}{i % 2}",
                                       defer_out=i + 1 < len(vit.blocks))
        if st is not None:
            st["layers"].append(sv)
    return stream


def _split3_weight(eng, p):
    """[hi | lo | hi] bf16 image of an fp32 weight (N, K) -> (N, 3 K), cached until the parameter changes."""
    key = ("split3", id(p))
    hit = eng._cache.get(key)
    stamp = (p._version, p.data_ptr())
    if hit is None or hit[0] != stamp:
        N, K = p.shape
        img = torch.empty(N, 3 * K, dtype=torch.bfloat16, device=p.device)
        L.check(L.split3_bf16(ops._p(p.detach()), K, ops._p(img), 3 * K, N, K, 1, 0, ops._stream()))
        hit = eng._cache[key] = (stamp, img)
    return hit[1]


SPLIT3_TAIL = os.environ.get("FOURM_VQ_SPLIT3", "1") != "0"


def _post_mlp_fwd(eng, vit, stream, R, st, prefix):
    """x.float() + fc2(tanh(fc1(norm_mlp(x.float()))))  with autocast DISABLED upstream (vit_models.py:494-496).
    Training (st given): fp32 operands on the fp32 matrix cores (fm_gemm_f32: v_mfma_f32_32x32x2_f32, exact fp32; the backward needs t).
    Inference: each fp32 operand as two bf16 halves, three-term products on the bf16 matrix cores (fm_split3_bf16 + one GEMM with K' = 3 K:
    ~1e-5 relative, 4 x the rate - round 5; FOURM_VQ_SPLIT3=0 keeps the exact form)."""
    ws, f32, D, Rp = eng.ws, torch.float32, eng.D, stream.shape[0]
    n = ws.get(prefix + ".n", (Rp, D), f32)
    mu = rs = None
    if st is not None:
        mu, rs = ws.get(prefix + ".mu", (Rp,), f32), ws.get(prefix + ".rs", (Rp,), f32)
    ops.layernorm_fwd(stream, vit.norm_mlp.weight, vit.norm_mlp.bias, n, mu, rs, eps=vit.norm_mlp.eps, R=R)
    hid = vit.post_mlp.fc1.weight.shape[0]
    out = ws.get(prefix + ".post", (Rp, D), f32)
    if st is None and SPLIT3_TAIL and D % 64 == 0 and hid % 64 == 0:
        bf = torch.bfloat16
        n3 = ws.get(prefix + ".n3", (Rp, 3 * D), bf)
        L.check(L.split3_bf16(ops._p(n), D, ops._p(n3), 3 * D, R, D, 0, 0, ops._stream()))
        pre = ws.get(prefix + ".pre", (Rp, hid), f32)
        ops.gemm_nt(n3, _split3_weight(eng, vit.post_mlp.fc1.weight), pre, epilogue=L.EPI_F32, bias=vit.post_mlp.fc1.bias, M=R, N=hid, K=3 * D)
        t3 = ws.get(prefix + ".t3", (Rp, 3 * hid), bf)
        L.check(L.split3_bf16(ops._p(pre), hid, ops._p(t3), 3 * hid, R, hid, 0, 1, ops._stream()))          # tanh, then the split
        ops.gemm_nt(t3, _split3_weight(eng, vit.post_mlp.fc2.weight), out, epilogue=L.EPI_F32, res=stream, bias=vit.post_mlp.fc2.bias, M=R, N=D, K=3 * hid)      # (EPI_F32 + res: no bf16 rounding of the branch)
        return out
    t = ws.get(prefix + ".t", (Rp, hid), f32)
    ops.gemm_nt(n, vit.post_mlp.fc1.weight.detach(), t, epilogue=L.EPI_TANH, bias=vit.post_mlp.fc1.bias, M=R, N=hid, K=D)
    ops.gemm_nt(t, vit.post_mlp.fc2.weight.detach(), out, epilogue=L.EPI_RESIDUAL, res=stream, bias=vit.post_mlp.fc2.bias, M=R, N=D, K=hid)
    if st is not None:
        st["post"] = dict(x=stream, n=n, t=t, mu=mu, rs=rs)
    return out


def _tokens(enc, x, st=None, prep=None):
    """(B, C, H, W) -> fp32 residual stream (Rp, D) after the last block (+ post MLP), plus (B, n_h, n_w).
    ``st`` (a dict) keeps what the backward needs (tokenizer training).  ``prep`` (VQ._prep): class maps (B, H, W) embedded by a table
    and / or a per-channel affine map, applied inside the patch gather."""
    eng = _engine(enc)
    ws, D = eng.ws, eng.D
    labels = prep is not None and prep["cls_emb"] is not None
    if labels:
        (B, Hh, Ww), C = x.shape, prep["cls_emb"].shape[1]
    else:
        B, C, Hh, Ww = x.shape
    P = enc.P_H if getattr(enc, "patch_proj", True) else 1          # (feature-map inputs: one token per position, vit_models.py:478-481)
    assert enc.P_H == enc.P_W and Hh % P == 0 and Ww % P == 0, f"Image sizes {Hh}x{Ww} must be divisible by patch size {P}"
    nh, nw = Hh // P, Ww // P
    G, R = nh * nw, B * nh * nw
    Rp = ru(R, 128)
    x = x.long().contiguous() if labels else x.float().contiguous()
    feat = C * P * P
    patches = ws.get("vq.patches", (Rp, ru(feat, 64)), torch.bfloat16)
    if prep is None:
        L.check(L.vq_patchify(ops._p(x), ops._p(patches), patches.stride(0), B, C, Hh, Ww, P, ops._stream()))
    else:
        emb = prep["cls_emb"].detach().float().contiguous() if labels else None
        L.check(L.vq_patchify_ex(None if labels else ops._p(x), ops._p(x) if labels else None, ops._p(emb), ops._p(prep["scale"]), ops._p(prep["shift"]),
                                 ops._p(patches), patches.stride(0), B, C, Hh, Ww, P, ops._stream()))
    pos = _pos_rows(eng, enc, B, nh, nw, Rp)
    stream = ws.get("vq.x0", (Rp, D), torch.float32)
    ops.gemm_nt(patches, eng.w(enc.proj.weight), stream, epilogue=L.EPI_RESIDUAL, res=pos, bias=enc.proj.bias, M=R, N=D, K=ru(feat, 64))
    if st is not None:
        st.update(layers=[], patches=patches, labels=x if labels else None, image=(B, C, Hh, Ww))
    stream = _blocks_fwd(eng, enc, stream, B, G, st, "enc" if st is not None else "vit")
    if hasattr(enc, "post_mlp"):
        stream = _post_mlp_fwd(eng, enc, stream, R, st, "vq")
    return eng, stream, (B, nh, nw)


@torch.no_grad()
def encoder_forward(enc, x):
    eng, stream, (B, nh, nw) = _tokens(enc, x)
    return stream[: B * nh * nw].view(B, nh, nw, eng.D).permute(0, 3, 1, 2).contiguous()


def _assign(vq, eng, z, R, G, B, nh, nw, want_quant):
    """Cosine-similarity nearest code of every latent row: tokens (B, nh, nw) int64 [+ quant (B, latent_dim, nh, nw) f32]."""
    ws, Ld = eng.ws, vq.latent_dim
    cb = vq.quantize._codebook
    K = cb.embed.shape[0]
    if getattr(cb, "euclidean", False):
        if not cb.is_initted():                             # kmeans_init=True: the first batch initialises the codebook (quantize_lucid.py:271)
            cb.init_embed_(z[:R], normalize=bool(vq.quantize.norm_latents))
        # Euclidean codebook (norm_codes=False): arg-max of <z, e> - |e|^2 / 2 over the raw codes; norm_latents normalises z first (:525-527)
        splits = max(1, min(16, K // 1024))
        wv = ws.get("vq.wv", (R, splits), torch.float32)
        wi = ws.get("vq.wi", (R, splits), torch.int32)
        tokens = torch.empty(B, nh, nw, dtype=torch.int64, device=z.device)
        quant = torch.empty(B, Ld, nh, nw, dtype=torch.float32, device=z.device) if want_quant else None
        L.check(L.vq_assign_bias(ops._p(z), z.stride(0), ops._p(cb.embed), ops._p(cb.code_bias()), ops._p(cb.embed), K, Ld, R, G,
                                 1 if vq.quantize.norm_latents else 0, ops._p(wv), ops._p(wi), splits, ops._p(tokens), ops._p(quant), ops._stream()))
        return (tokens, quant) if want_quant else tokens
    if not cb.is_initted():                                 # kmeans_init=True: the first batch initialises the codebook (quantize_lucid.py:394)
        cb.init_embed_(z[:R])
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
    tokens = torch.empty(B, nh, nw, dtype=torch.int64, device=z.device)
    quant = torch.empty(B, Ld, nh, nw, dtype=torch.float32, device=z.device) if want_quant else None
    # cosine similarity normalises the latents (quantize_lucid.py:394-395); norm_latents only moves that
    # normalisation in front of the (training-time) commitment loss
    L.check(L.vq_assign(ops._p(z), z.stride(0), ops._p(en), ops._p(cb.embed), K, Ld, R, G, 1, ops._p(wv), ops._p(wi), splits,
                        ops._p(tokens), ops._p(quant), ops._stream()))
    return (tokens, quant) if want_quant else tokens


@torch.no_grad()
def vq_encode(vq, x):
    enc = vq.encoder
    eng, stream, (B, nh, nw) = _tokens(enc, x, prep=vq._prep())
    ws, D, Ld = eng.ws, eng.D, vq.latent_dim
    G, R = nh * nw, B * nh * nw
    # 1x1 convolution to the latent dimension: fp32 like the codebook search that follows (the tokenization script runs without
    # autocast, save_vq_tokens.py; 0.2 % of the FLOPs)
    z = ws.get("vq.z", (stream.shape[0], Ld), torch.float32)
    ops.gemm_nt(stream, vq.quant_proj.weight.detach().reshape(Ld, D), z, epilogue=L.EPI_F32, bias=vq.quant_proj.bias, M=R, N=Ld, K=D)
    tokens, quant = _assign(vq, eng, z, R, G, B, nh, nw, True)
    vq._last_latents = z[:R].view(B, G, Ld)
    return quant, torch.zeros(1, device=x.device), tokens


# ---------------------------------------------------------------------------------------------------------------------------------
# detokenizer (ViTDecoder) and the gradient path of tokenizer training
#   upstream: VQVAE.decode_quant / forward (vq/vqvae.py:454-481), ViTDecoder.forward (vq/models/vit_models.py:617-648),
#             VectorQuantize.forward training branch (vq/quantizers/quantize_lucid.py:533-541)
# ---------------------------------------------------------------------------------------------------------------------------------
def _quant_rows(vq, tokens, dev):
    """embed[tokens] as GEMM operand rows (R, latent_dim) f32."""
    cb = vq.quantize._codebook
    t = tokens.reshape(-1).contiguous()
    rows = torch.empty(t.numel(), cb.embed.shape[1], dtype=torch.float32, device=dev)
    L.check(L.embed_rows_f32(ops._p(cb.embed), ops._p(t), ops._p(rows), rows.stride(0), t.numel(), cb.embed.shape[1], ops._stream()))
    return rows


def _decode_rows(vq, q_rows, B, nh, nw, st=None):
    """Quantised rows (R, latent_dim) f32 -> image (B, C, H, W) f32 through post_quant_proj, the decoder blocks and out_proj."""
    dec = vq.decoder
    if hasattr(dec, "out_conv"):
        raise NotImplementedError("ViTDecoder with out_conv=True has no HIP path")
    eng = _engine(dec)
    ws, D, Ld = eng.ws, eng.D, vq.latent_dim
    G, R = nh * nw, B * nh * nw
    Rp = ru(R, 128)
    f32 = torch.float32
    pos = _pos_rows(eng, dec, B, nh, nw, Rp)
    stream = ws.get("dec.x0", (Rp, D), f32)
    pq = vq.post_quant_proj
    # 1x1 convolution latent_dim -> decoder width, fp32 like the quantizer in front of it (K = 32: 0.1 % of the FLOPs), + position table
    ops.gemm_nt(q_rows, pq.weight.detach().reshape(D, Ld), stream, epilogue=L.EPI_RESIDUAL, res=pos, bias=pq.bias, M=R, N=D, K=Ld)
    if st is not None:
        st.update(layers=[], q_rows=q_rows)
    stream = _blocks_fwd(eng, dec, stream, B, G, st, "dec" if st is not None else "vit")
    if hasattr(dec, "post_mlp"):
        stream = _post_mlp_fwd(eng, dec, stream, R, st, "decpost")
    P, C = (dec.P_H if dec.patch_proj else 1), dec.out_channels
    Fo = C * P * P
    xb = ws.get("dec.xb", (Rp, D), torch.bfloat16)                      # out_proj runs under autocast: bf16 operands, fp32 result
    ops.f32_to_bf16(stream, xb)
    rows = ws.get("dec.rows", (Rp, Fo), f32)
    ops.gemm_nt(xb, eng.w(dec.out_proj.weight), rows, epilogue=L.EPI_F32, bias=dec.out_proj.bias, M=R, N=Fo, K=D)
    img = torch.empty(B, C, nh * P, nw * P, dtype=f32, device=q_rows.device)
    L.check(L.vq_unpatchify(ops._p(rows), rows.stride(0), ops._p(img), B, C, nh * P, nw * P, P, ops._stream()))
    if st is not None:
        st.update(xb=xb, x_final=stream)
    return img


@torch.no_grad()
def vqvae_decode_quant(vq, quant):
    """(B, latent_dim, h, w) -> (B, C, H, W)   [vqvae.py:454-465]"""
    B, Ld, nh, nw = quant.shape
    rows = quant.detach().float().permute(0, 2, 3, 1).reshape(B * nh * nw, Ld).contiguous()
    return _decode_rows(vq, rows, B, nh, nw)


@torch.no_grad()
def vqvae_decode_tokens(vq, tokens):
    B, nh, nw = tokens.shape
    return _decode_rows(vq, _quant_rows(vq, tokens, tokens.device), B, nh, nw)


def _post_mlp_bwd(eng, vit, sp, g, g_bf, R):
    """Backward of x + fc2(tanh(fc1(norm_mlp(x)))) in fp32.  g (Rp, D) f32: in = d(out), out = d(x); g_bf receives the bf16 copy.
    Every GEMM runs on the fp32 matrix cores (fm_gemm_f32's MFMA kernel wants both operands contiguous along the reduction): dX against
    transposed fp32 copies of the two weights, dW = dY^T X as an NT product of the transposed row blocks (layout copies, a few MB)."""
    ws, f32, D = eng.ws, torch.float32, eng.D
    fc1, fc2 = vit.post_mlp.fc1, vit.post_mlp.fc2
    hid = fc1.weight.shape[0]
    gT = g[:R].t().contiguous()                                          # (D, R)
    if fc2.weight.requires_grad:
        tT = sp["t"][:R].t().contiguous()                                # (hid, R)
        ops._gemm_f32(gT, tT, eng.grad_view(fc2.weight), M=D, N=hid, K=R, accumulate=True)        # dW2[d][h] += sum_r g[r][d] t[r][h]
        ops.colsum(g, eng.grad_view(fc2.bias), D, R=R)
        del tT
    dt = ws.get("bwd.post.dt", tuple(sp["t"].shape), f32)
    ops.gemm_nt(g, fc2.weight.detach().t().contiguous(), dt, epilogue=L.EPI_F32, M=R, N=hid, K=D)
    L.check(L.tanh_bwd_f32(ops._p(dt), ops._p(sp["t"]), ops._p(dt), R, hid, dt.stride(0), ops._stream()))
    if fc1.weight.requires_grad:
        dtT, nT = dt[:R].t().contiguous(), sp["n"][:R].t().contiguous()  # (hid, R), (D, R)
        ops._gemm_f32(dtT, nT, eng.grad_view(fc1.weight), M=hid, N=D, K=R, accumulate=True)
        ops.colsum(dt, eng.grad_view(fc1.bias), hid, R=R)
        del dtT, nT
    dn = ws.get("bwd.post.dn", tuple(g.shape), f32)
    ops.gemm_nt(dt, fc1.weight.detach().t().contiguous(), dn, epilogue=L.EPI_F32, M=R, N=D, K=hid)
    nm = vit.norm_mlp
    # (the fp32 LayerNorm backward writes its second copy in fp32 too - it serves the verification mode: convert separately)
    ops.layernorm_bwd(dn, sp["x"], nm.weight, sp["mu"], sp["rs"], g, dres=g, dw=eng._g(nm.weight), db=eng._g(nm.bias), R=R)
    ops.f32_to_bf16(g, g_bf)


def _blocks_bwd(eng, vit, st, g, g_bf, B, G):
    none = dict(mask_kind=L.MASK_NONE)
    for i in reversed(range(len(vit.blocks))):
        eng.encoder_block_bwd(vit.blocks[i], st["layers"][i], g, g_bf, B, G, none)


def vqvae_train_forward(vq, x):
    """Training forward of the VQ-VAE: returns (dec (B, C, H, W) f32, code_loss (1,) f32, saved state)."""
    enc = vq.encoder
    trainable_enc = any(p.requires_grad for p in enc.parameters()) or vq.quant_proj.weight.requires_grad
    st = dict(enc={} if trainable_enc else None, dec={})
    if enc.pos_emb.requires_grad or vq.decoder.pos_emb.requires_grad:
        raise NotImplementedError("learnable position embeddings have no gradient kernel (learnable_pos_emb=False upstream default)")
    eng, stream, (B, nh, nw) = _tokens(enc, x, st["enc"], prep=vq._prep())
    ws, D, Ld = eng.ws, eng.D, vq.latent_dim
    G, R = nh * nw, B * nh * nw
    z = ws.get("vq.z", (stream.shape[0], Ld), torch.float32)
    ops.gemm_nt(stream, vq.quant_proj.weight.detach().reshape(Ld, D), z, epilogue=L.EPI_F32, bias=vq.quant_proj.bias, M=R, N=Ld, K=D)
    lat_grad = L.vq_latent_grad_normalized if vq.quantize.norm_latents else L.vq_latent_grad      # (norm_latents: x = l2norm(z), :525-527)
    tokens = _assign(vq, eng, z, R, G, B, nh, nw, None)
    q_rows = _quant_rows(vq, tokens, x.device)
    cw = float(vq.quantize.commitment_weight)
    code_loss = torch.zeros(1, dtype=torch.float32, device=x.device)
    cb = vq.quantize._codebook
    if cw > 0:
        L.check(lat_grad(ops._p(z), z.stride(0), ops._p(cb.embed), ops._p(tokens.reshape(-1)), None, 0, None, cw, None, 0,
                         ops._p(code_loss), R, Ld, ops._stream()))
    dec = _decode_rows(vq, q_rows, B, nh, nw, st["dec"])
    st.update(z=z, tokens=tokens, x_enc_final=stream, dims=(B, nh, nw), embed_at_forward=cb.embed.clone() if vq.quantize.training else cb.embed)
    if vq.quantize.training:
        if getattr(cb, "euclidean", False):                # (the Euclidean codebook sees l2norm(z) when norm_latents: quantize_lucid.py:525-527)
            cb.ema_update_(z[:R], tokens, normalize=bool(vq.quantize.norm_latents))
        else:
            cb.ema_update_(z[:R], tokens)                  # after the code assignment, as upstream (quantize_lucid.py:409-426)
    return dec, code_loss, st


def vqvae_train_backward(vq, st, g_dec, g_loss):
    """d(objective)/d(dec) (B, C, H, W) and d(objective)/d(code_loss) -> parameter gradients (accumulated, then attached as .grad)."""
    dec, enc = vq.decoder, vq.encoder
    B, nh, nw = st["dims"]
    G, R = nh * nw, B * nh * nw
    de = _engine(dec)
    de.open_window(list(dec.parameters()) + list(vq.post_quant_proj.parameters()))
    ws, D, Ld, f32, bf = de.ws, de.D, vq.latent_dim, torch.float32, torch.bfloat16
    Rp = ru(R, 128)
    sd = st["dec"]
    P, C = (dec.P_H if dec.patch_proj else 1), dec.out_channels
    Fo = C * P * P
    # ---- out_proj ------------------------------------------------------------------------------------------------------------------
    drow = ws.get("bwd.drow", (Rp, ru(Fo, 64)), bf)
    gd = (g_dec if g_dec is not None else torch.zeros(B, C, nh * P, nw * P, device=sd["xb"].device)).float().contiguous()
    L.check(L.vq_patchify(ops._p(gd), ops._p(drow), drow.stride(0), B, C, nh * P, nw * P, P, ops._stream()))
    de._dW(drow, sd["xb"], dec.out_proj, R, n_cols=Fo)
    g_bf = ws.get("bwd.g_bf", (Rp, D), bf)
    g = ws.get("bwd.g", (Rp, D), f32)
    ops.gemm_nt(drow, de.wt(dec.out_proj.weight), g_bf, M=R, N=D, K=ru(Fo, 64))
    ops.bf16_to_f32_scaled(g_bf, g)
    if hasattr(dec, "post_mlp"):
        _post_mlp_bwd(de, dec, sd["post"], g, g_bf, R)
    _blocks_bwd(de, dec, sd, g, g_bf, B, G)
    # ---- post_quant_proj (fp32) --------------------------------------------------------------------------------------------------------
    pq = vq.post_quant_proj
    if pq.weight.requires_grad:
        ops.gemm_tn(g, sd["q_rows"], de.grad_view(pq.weight).view(D, Ld), N=D, K=Ld, R=R)
        ops.colsum(g, de.grad_view(pq.bias), D, R=R)
    de.attach(list(dec.parameters()) + list(pq.parameters()))
    se = st["enc"]
    if se is None:
        return
    ee = _engine(enc)
    ee.open_window(list(enc.parameters()) + list(vq.quant_proj.parameters()))
    ews, De = ee.ws, ee.D
    dq = ws.get("bwd.dq", (Rp, Ld), f32)
    ops.gemm_nt(g, pq.weight.detach().reshape(D, Ld).t(), dq, epilogue=L.EPI_F32, M=R, N=Ld, K=D)
    # ---- quantizer: straight-through + commitment term (against the codebook the forward used) ---------------------------------------
    z, tokens = st["z"], st["tokens"]
    dz = ews.get("bwd.dz", tuple(z.shape), f32)
    gl = None if g_loss is None else g_loss.reshape(1).float().contiguous()
    lat_grad = L.vq_latent_grad_normalized if vq.quantize.norm_latents else L.vq_latent_grad
    L.check(lat_grad(ops._p(z), z.stride(0), ops._p(st["embed_at_forward"]), ops._p(tokens.reshape(-1)), ops._p(dq), dq.stride(0), ops._p(gl),
                     float(vq.quantize.commitment_weight), ops._p(dz), dz.stride(0), None, R, Ld, ops._stream()))
    # ---- quant_proj (fp32) -------------------------------------------------------------------------------------------------------------
    qp = vq.quant_proj
    xf = st["x_enc_final"]
    if qp.weight.requires_grad:
        ops.gemm_tn(dz, xf, ee.grad_view(qp.weight).view(Ld, De), N=Ld, K=De, R=R)
        ops.colsum(dz, ee.grad_view(qp.bias), Ld, R=R)
    ge = ews.get("bwd.g", (xf.shape[0], De), f32)
    ge_bf = ews.get("bwd.g_bf", (xf.shape[0], De), bf)
    ops.gemm_nt(dz, qp.weight.detach().reshape(Ld, De).t(), ge, epilogue=L.EPI_F32, M=R, N=De, K=Ld)
    if hasattr(enc, "post_mlp"):
        _post_mlp_bwd(ee, enc, se["post"], ge, ge_bf, R)
    else:
        ops.f32_to_bf16(ge, ge_bf)
    _blocks_bwd(ee, enc, se, ge, ge_bf, B, G)
    ee._dW(ge_bf, se["patches"], enc.proj, R)
    extra = []
    if se["labels"] is not None and vq.cls_emb.weight.requires_grad:       # the class-embedding table behind the patch projection
        Bi, Ci, Hi, Wi = se["image"]
        P_ = enc.P_H if enc.patch_proj else 1
        dpatch = ews.get("bwd.dpatch", tuple(se["patches"].shape), bf)
        ops.gemm_nt(ge_bf, ee.wt(enc.proj.weight), dpatch, M=R, N=Ci * P_ * P_, K=De)
        ee.open_window([vq.cls_emb.weight])
        L.check(L.vq_cls_emb_bwd(ops._p(dpatch), dpatch.stride(0), ops._p(se["labels"]), ops._p(ee.grad_view(vq.cls_emb.weight)), Bi, Ci, Hi, Wi, P_,
                                 ops._stream()))
        extra = [vq.cls_emb.weight]
    ee.attach(list(enc.parameters()) + list(qp.parameters()) + extra)


class VQVAEStep(torch.autograd.Function):
    """Bridges the hand-written backward into autograd: ``dec, code_loss = VQVAE.forward(x)`` are leaves of the caller's loss graph."""

    @staticmethod
    def forward(ctx, anchor, vq, x):
        dec, code_loss, st = vqvae_train_forward(vq, x)
        ctx.vq, ctx.st = vq, st
        return dec, code_loss

    @staticmethod
    def backward(ctx, g_dec, g_loss):
        vqvae_train_backward(ctx.vq, ctx.st, g_dec, g_loss)
        ctx.st = None
        return None, None, None
