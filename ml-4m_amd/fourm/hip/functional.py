"""Stand-alone (inference, no autograd) entry points for pieces of the trunk: what upstream code reaches
by calling ``norm(x)``, ``blk(x, mask)`` or ``to_logits(x)`` directly.  Each call runs HIP kernels on
scratch buffers; results are returned as new fp32 tensors like the upstream modules do without autocast."""
import torch

from . import _lib as L
from . import ops


def _no_grad_only(what):
    if torch.is_grad_enabled():
        raise NotImplementedError(f"{what}: differentiable stand-alone calls are not implemented; train through FourM.forward")


def _bf16_rows(x2d):
    """Zero-padded bf16 copy (rows to 128, columns to 64): the layout contract of the GEMM kernels."""
    R, K = x2d.shape
    buf = torch.zeros(ops.ru(R, 128), ops.ru(K, 64), dtype=torch.bfloat16, device=x2d.device)
    buf[:R, :K] = x2d.to(torch.bfloat16)
    return buf


def layer_norm(x, weight, bias, eps):
    _no_grad_only("LayerNorm")
    shp = x.shape
    x2 = x.reshape(-1, shp[-1]).float().contiguous()
    y = torch.empty_like(x2)
    ops.layernorm_fwd(x2, weight, bias, y, eps=eps)
    return y.reshape(shp).to(x.dtype)


def linear(x, weight, bias, out_dtype=None):
    """y = x W^T + b with bf16 operands / fp32 accumulation (the autocast semantics of the trunk)."""
    _no_grad_only("linear")
    from fourm.hip.engine import ru
    shp = x.shape
    x2 = x.reshape(-1, shp[-1])
    R, K = x2.shape
    N = weight.shape[0]
    if R == 0:                       # e.g. y[mod_mask == id] of a modality without rows (fm.py:535-541 returns (0, V) too)
        return x2.new_zeros((*shp[:-1], N), dtype=out_dtype or x.dtype)
    xb = _bf16_rows(x2)
    wb = torch.zeros(N, ru(K, 64), dtype=torch.bfloat16, device=x.device)
    ops.cast_pad(weight.detach().float(), wb)
    out = torch.empty(R, ru(N, 4), dtype=torch.bfloat16, device=x.device)
    ops.gemm_nt(xb, wb, out, bias=bias, M=R, N=N, K=ru(K, 64))
    y = out[:, :N]
    return y.reshape(*shp[:-1], N).to(out_dtype or (x.dtype if x.dtype != torch.float32 else torch.float32))


def embed_modality(emb, d, is_dec: bool):
    """A modality embedding's own ``forward(d)`` / ``forward_embed(d)``: token rows ``x`` and ``emb`` = position +
    modality embedding for EVERY position, nothing zeroed (encoder_embeddings.py:87-121,184-211,280-309,387-421,
    decoder_embeddings.py:98-139,226-255).  One launch of the selection kernel in its raw view (+ the projection
    GEMM for pixels / dense embeddings).  Returns (x, emb) as fp32 (B, L, D)."""
    _no_grad_only(type(emb).__name__)
    from fourm.hip.engine import fill_mod_desc, ru
    t = d["tensor"]
    B, dev, D = t.shape[0], t.device, emb.dim_tokens
    desc = L.SelectDesc()
    keep = fill_mod_desc(desc.mods[0], d, emb, is_dec, 0, 0, raw=2, name=type(emb).__name__)
    Lm = desc.mods[0].L
    R, Rp = B * Lm, ru(B * Lm, 128)
    f32, i32 = torch.float32, torch.int32
    tokens, e = torch.zeros(Rp, D, dtype=f32, device=dev), torch.zeros(Rp, D, dtype=f32, device=dev)
    side = dict(mask=torch.empty(B, Lm, dtype=torch.bool, device=dev), mod=torch.empty(B, Lm, dtype=torch.int16, device=dev),
                smod=torch.empty(B, Lm, dtype=i32, device=dev), ssrc=torch.empty(B, Lm, dtype=i32, device=dev),
                spos=torch.empty(B, Lm, dtype=i32, device=dev))
    desc.n_mods, desc.batch, desc.dim, desc.n_keep, desc.n_reg, desc.total_len = 1, B, D, Lm, 0, Lm
    desc.is_decoder, desc.raw = 1 if is_dec else 0, 2
    desc.tokens, desc.emb = tokens.data_ptr(), e.data_ptr()
    desc.out_mask, desc.out_mod = side["mask"].data_ptr(), side["mod"].data_ptr()
    desc.slot_mod, desc.slot_src, desc.slot_pos = side["smod"].data_ptr(), side["ssrc"].data_ptr(), side["spos"].data_ptr()
    if is_dec:
        side.update(tgt=torch.empty(B, Lm, dtype=torch.int64, device=dev), cs=torch.empty(B, Lm, dtype=i32, device=dev),
                    pre=torch.empty(B, Lm, dtype=torch.int16, device=dev), hidx=torch.empty(B, Lm, dtype=i32, device=dev),
                    mtok=torch.zeros(D, dtype=f32, device=dev))
        desc.target_ids, desc.out_cs = side["tgt"].data_ptr(), side["cs"].data_ptr()
        desc.out_mod_pre, desc.out_mod_index, desc.mask_token = side["pre"].data_ptr(), side["hidx"].data_ptr(), side["mtok"].data_ptr()
    rows = weight = None
    if emb.kind == L.KIND_PATCH:
        weight = emb.proj.weight
        rows = torch.zeros(Rp, ru(weight.shape[1], 64), dtype=torch.bfloat16, device=dev)
        desc.patch_rows, desc.patch_ld = rows.data_ptr(), rows.shape[1]
    elif emb.kind == L.KIND_SEQ_EMB:
        weight = emb.emb_proj.weight
        rows = torch.zeros(Rp, ru(emb.orig_emb_dim, 64), dtype=torch.bfloat16, device=dev)
        desc.seqemb_rows, desc.seqemb_ld = rows.data_ptr(), rows.shape[1]
    L.check(L.select_embed(ops.C.byref(desc), ops._stream()))
    if rows is not None:      # x = proj(patches) / emb_proj(embeddings) (+ bias, already in the token rows)
        wb = torch.zeros(weight.shape[0], rows.shape[1], dtype=torch.bfloat16, device=dev)
        ops.cast_pad(weight.detach().float(), wb)
        ops.gemm_nt(rows, wb, tokens, epilogue=L.EPI_RESIDUAL, res=tokens, M=R, N=D)
    del keep
    return tokens[:R].view(B, Lm, D), e[:R].view(B, Lm, D)


# block module -> weak reference to the FourM that owns it (filled by FourM.__init__ / FourM.engine; a registry rather than an
# attribute on the block so that pickling / deepcopy of the modules never meets a weakref)
import weakref
_BLOCK_OWNER = weakref.WeakKeyDictionary()


def register_blocks(model):
    ref = weakref.ref(model)
    for blk in list(model.encoder) + list(model.decoder):
        _BLOCK_OWNER[blk] = ref


def _engine_of(block):
    ref = _BLOCK_OWNER.get(block)
    model = ref() if ref is not None else None
    if model is None:
        raise RuntimeError("this block is not attached to a live FourM model (blocks compute through the model's engine; "
                           "after copy.deepcopy / unpickling touch model.engine once)")
    return model.engine


def _mask_args(mask, B, Nq, Nk):
    if mask is None:
        return dict(mask_kind=L.MASK_NONE)
    m = mask.bool()
    if m.dim() == 3 and m.shape[1] == 1:
        return dict(mask_kind=L.MASK_KEYPAD, kpad=m.reshape(B, Nk).contiguous())
    if m.dim() == 2:
        m = m[None].expand(B, Nq, Nk)
    return dict(mask_kind=L.MASK_DENSE, dense=m.expand(B, Nq, Nk).contiguous())


def encoder_block(block, x, mask):
    _no_grad_only("Block")
    eng = _engine_of(block)
    eng.prepare()
    B, N, D = x.shape
    xin = eng.ws.get("fn.x", (ops.ru(B * N, 128), D), torch.float32)
    xin[: B * N] = x.reshape(B * N, D).float()
    out = eng.encoder_block_fwd(block, xin, B, N, _mask_args(mask, B, N, N), None, "enc0")
    return out[: B * N].reshape(B, N, D).clone()


def decoder_block(block, x, context, sa_mask, xa_mask):
    _no_grad_only("DecoderBlock")
    eng = _engine_of(block)
    eng.prepare()
    B, M, D = x.shape
    N = context.shape[1]
    xin = eng.ws.get("fn.y", (ops.ru(B * M, 128), D), torch.float32)
    cin = eng.ws.get("fn.c", (ops.ru(B * N, 128), D), torch.float32)
    xin[: B * M] = x.reshape(B * M, D).float()
    cin[: B * N] = context.reshape(B * N, D).float()
    out = eng.decoder_block_fwd(block, xin, cin, B, M, N, _mask_args(sa_mask, B, M, M), _mask_args(xa_mask, B, M, N), None, "dec0")
    return out[: B * M].reshape(B, M, D).clone()
