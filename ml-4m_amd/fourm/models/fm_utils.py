"""Parameter containers for the 4M trunk.

These modules own the parameters under the same names (and therefore the same state_dict keys) as
upstream's ``fourm/models/fm_utils.py`` — ``norm1.weight``, ``attn.qkv.weight``, ``mlp.fc3.weight`` …
— but they do not compute: the arithmetic of a block lives in ``fourm.hip.engine`` and runs as HIP
kernels.  Calling a block directly routes through the same engine (inference only).
"""
import math

import torch
import torch.nn as nn


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def build_1d_sincos_posemb(max_len, embed_dim=1024, temperature=10000.):
    """(1, max_len, embed_dim): sin half then cos half.  [upstream fm_utils.py:32-44]"""
    if embed_dim % 2:
        raise AssertionError("Embed dimension must be divisible by 2 for 1D sin-cos position embedding")
    half = embed_dim // 2
    freq = 1. / (temperature ** (torch.arange(half, dtype=torch.float32) / half))
    ang = torch.arange(max_len, dtype=torch.float32)[:, None] * freq[None]
    return torch.cat([ang.sin(), ang.cos()], dim=1)[None]


def build_2d_sincos_posemb(h, w, embed_dim=1024, temperature=10000.0):
    """(1, h*w, embed_dim).  Positions are enumerated with the first ('w') grid coordinate slow and that
    coordinate feeds the first half of the channels — the upstream convention [fm_utils.py:46-61], kept
    because released checkpoints carry these tables as buffers."""
    if embed_dim % 4:
        raise AssertionError("Embed dimension must be divisible by 4 for 2D sin-cos position embedding")
    q = embed_dim // 4
    freq = 1. / (temperature ** (torch.arange(q, dtype=torch.float32) / q))
    slow = torch.arange(w, dtype=torch.float32)[:, None].expand(w, h).reshape(-1)
    fast = torch.arange(h, dtype=torch.float32)[None, :].expand(w, h).reshape(-1)
    a, b = slow[:, None] * freq[None], fast[:, None] * freq[None]
    return torch.cat([a.sin(), a.cos(), b.sin(), b.cos()], dim=1)[None]


def _engine():
    from fourm.hip import functional
    return functional


class LayerNorm(nn.Module):
    """LayerNorm whose bias can be switched off; the bias then stays in the state_dict as an all-zero
    *buffer* (checkpoint compatibility with upstream fm_utils.py:93-108)."""

    def __init__(self, normalized_shape: int, eps=1e-5, bias=True):
        super().__init__()
        self.eps = eps
        self.normalized_shape = (normalized_shape,)
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        if bias:
            self.bias = nn.Parameter(torch.zeros(normalized_shape))
        else:
            self.register_buffer("bias", torch.zeros(normalized_shape))

    def forward(self, x):
        return _engine().layer_norm(x, self.weight, self.bias, self.eps)


def make_norm(norm_layer, width):
    """Instantiate the user's norm_layer and make sure it is one the kernels implement."""
    m = norm_layer(width)
    if not isinstance(m, (LayerNorm, nn.LayerNorm)):
        raise TypeError(f"norm_layer {type(m).__name__} has no HIP implementation (LayerNorm only)")
    if isinstance(m, nn.LayerNorm) and not m.elementwise_affine:
        raise TypeError("LayerNorm without affine parameters is not supported")
    return m


class Mlp(nn.Module):
    """fc2(act(fc1 x))   [upstream fm_utils.py:111-126]"""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0., bias=True):
        super().__init__()
        if drop:
            raise NotImplementedError("dropout inside the MLP is not implemented in the HIP path")
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features or in_features, bias=bias)
        self.hidden_features = hidden_features


class GatedMlp(nn.Module):
    """fc2(act(fc1 x) * fc3 x) with hidden = int(2/3 * hidden)   [upstream fm_utils.py:129-144]"""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.SiLU, bias=True):
        super().__init__()
        hidden = int(2 * (hidden_features or in_features) / 3)
        self.fc1 = nn.Linear(in_features, hidden, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden, out_features or in_features, bias=bias)
        self.fc3 = nn.Linear(in_features, hidden, bias=bias)
        self.hidden_features = hidden


def _check_attn_args(attn_drop, proj_drop, allow_zero_attn):
    if attn_drop or proj_drop:
        raise NotImplementedError("attention / projection dropout is not implemented in the HIP path")


class Attention(nn.Module):
    """Self-attention parameters: fused qkv (rows ordered q | k | v) + proj   [upstream fm_utils.py:147-180]"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True, attn_drop=0., proj_drop=0., allow_zero_attn=False):
        super().__init__()
        _check_attn_args(attn_drop, proj_drop, allow_zero_attn)
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.allow_zero_attn = allow_zero_attn          # softmax1 (upstream fm_utils.py:28-30): fm_attn_args.zero_attn
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)


class CrossAttention(nn.Module):
    """Cross-attention parameters: q on the queries, fused kv (rows k | v) on the context + proj
    [upstream fm_utils.py:182-219]"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True, attn_drop=0., proj_drop=0., allow_zero_attn=False):
        super().__init__()
        _check_attn_args(attn_drop, proj_drop, allow_zero_attn)
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.allow_zero_attn = allow_zero_attn
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim, bias=proj_bias)


class NormAttention(Attention):
    """+ per-head LayerNorm on q and k (QK-norm)   [upstream fm_utils.py:222-261]"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True, norm_layer=nn.LayerNorm, **kw):
        super().__init__(dim, num_heads, qkv_bias, proj_bias, **kw)
        self.q_norm = norm_layer(dim // num_heads)
        self.k_norm = norm_layer(dim // num_heads)


class NormCrossAttention(CrossAttention):
    """[upstream fm_utils.py:264-307]"""

    def __init__(self, dim, num_heads=8, qkv_bias=False, proj_bias=True, norm_layer=nn.LayerNorm, **kw):
        super().__init__(dim, num_heads, qkv_bias, proj_bias, **kw)
        self.q_norm = norm_layer(dim // num_heads)
        self.k_norm = norm_layer(dim // num_heads)


def _mlp(dim, mlp_ratio, act_layer, mlp_bias, gated_mlp, drop):
    hidden = int(dim * mlp_ratio)
    if gated_mlp:
        return GatedMlp(in_features=dim, hidden_features=hidden, act_layer=act_layer, bias=mlp_bias)
    return Mlp(in_features=dim, hidden_features=hidden, act_layer=act_layer, bias=mlp_bias, drop=drop)


class DropPath(nn.Module):
    """Stochastic depth per sample (upstream fm_utils.py:64-87): in training mode a residual branch's output is multiplied by
    floor(keep_prob + u) / keep_prob with one uniform u per sample.  Parameter-free marker: the engine draws the scales
    (``sample_scale``) and applies them with fm_scale_rows_bf16 in the forward and in the backward."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def sample_scale(self, batch: int, device, uniforms=None):
        keep = 1.0 - self.drop_prob
        u = torch.rand(batch, device=device) if uniforms is None else uniforms.to(device=device, dtype=torch.float32)
        return (torch.floor(keep + u) / keep).float().contiguous()

    def extra_repr(self) -> str:
        return "p={}".format(self.drop_prob)


def _no_drop_path(rate):
    return DropPath(rate) if rate and rate > 0. else nn.Identity()


class Block(nn.Module):
    """Pre-norm encoder block: x + attn(norm1 x); x + mlp(norm2 x)   [upstream fm_utils.py:310-334]"""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=True, proj_bias=True, mlp_bias=True, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, gated_mlp=False, qk_norm=False, allow_zero_attn=False):
        super().__init__()
        self.norm1 = make_norm(norm_layer, dim)
        cls = NormAttention if qk_norm else Attention
        extra = dict(norm_layer=norm_layer) if qk_norm else {}
        self.attn = cls(dim, num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias, attn_drop=attn_drop, proj_drop=drop,
                        allow_zero_attn=allow_zero_attn, **extra)
        self.drop_path = _no_drop_path(drop_path)
        self.norm2 = make_norm(norm_layer, dim)
        self.mlp = _mlp(dim, mlp_ratio, act_layer, mlp_bias, gated_mlp, drop)

    def forward(self, x, mask=None):
        return _engine().encoder_block(self, x, mask)


class DecoderBlock(nn.Module):
    """Pre-norm decoder block: self-attention, cross-attention on the (normed) context, MLP
    [upstream fm_utils.py:337-366]"""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=True, proj_bias=True, mlp_bias=True, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, gated_mlp=False, qk_norm=False, allow_zero_attn=False):
        super().__init__()
        self.norm1 = make_norm(norm_layer, dim)
        extra = dict(norm_layer=norm_layer) if qk_norm else {}
        sa, xa = (NormAttention, NormCrossAttention) if qk_norm else (Attention, CrossAttention)
        kw = dict(num_heads=num_heads, qkv_bias=qkv_bias, proj_bias=proj_bias, attn_drop=attn_drop, proj_drop=drop,
                  allow_zero_attn=allow_zero_attn, **extra)
        self.self_attn = sa(dim, **kw)
        self.cross_attn = xa(dim, **kw)
        self.query_norm = make_norm(norm_layer, dim)
        self.context_norm = make_norm(norm_layer, dim)
        self.drop_path = _no_drop_path(drop_path)
        self.norm2 = make_norm(norm_layer, dim)
        self.mlp = _mlp(dim, mlp_ratio, act_layer, mlp_bias, gated_mlp, drop)

    def forward(self, x, context, sa_mask=None, xa_mask=None):
        return _engine().decoder_block(self, x, context, sa_mask, xa_mask)


def act_name(act_module: nn.Module) -> str:
    if isinstance(act_module, nn.SiLU):
        return "silu"
    if isinstance(act_module, nn.GELU) and getattr(act_module, "approximate", "none") == "none":
        return "gelu"
    raise NotImplementedError(f"activation {type(act_module).__name__} has no HIP epilogue (SiLU-gated or exact GELU only)")


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
