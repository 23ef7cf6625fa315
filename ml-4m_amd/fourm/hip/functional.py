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
    xb = _bf16_rows(x2)
    wb = torch.zeros(N, ru(K, 64), dtype=torch.bfloat16, device=x.device)
    ops.cast_pad(weight.detach().float(), wb)
    out = torch.empty(R, ru(N, 4), dtype=torch.bfloat16, device=x.device)
    ops.gemm_nt(xb, wb, out, bias=bias, M=R, N=N, K=ru(K, 64))
    y = out[:, :N]
    return y.reshape(*shp[:-1], N).to(out_dtype or (x.dtype if x.dtype != torch.float32 else torch.float32))


def _engine_of(block):
    eng = getattr(block, "_fourm_engine", None)
    if eng is None:
        raise RuntimeError("this block is not attached to a FourM model (blocks compute through the model's engine)")
    return eng()


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
