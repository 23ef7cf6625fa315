"""ViT tokenizer encoder / decoder: parameter owners in the upstream layout (``fourm/vq/models/vit_models.py``:
``ViTEncoder`` :338-501, ``ViTDecoder`` :504-659, ``Block`` :232-246, ``Attention`` :165-197, ``Mlp`` :145-162, factories :664-859).
The arithmetic runs in ``fourm.vq.engine`` on the same HIP kernels as the 4M trunk (patch-projection GEMM,
LayerNorm, bias+GELU MLP, unmasked attention)."""
import math
from functools import partial
from typing import Optional

import torch
import torch.nn as nn


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


def build_2d_sincos_posemb(h, w, embed_dim=1024, temperature=10000.):
    """(1, embed_dim, h, w) table, upstream convention (vit_models.py:38-52)."""
    from fourm.models.fm_utils import build_2d_sincos_posemb as flat
    return flat(h, w, embed_dim, temperature)[0].reshape(h, w, embed_dim).permute(2, 0, 1)[None].contiguous()


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.hidden_features = hidden_features or in_features


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError("dropout is not implemented in the HIP tokenizer path")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        if drop or drop_path:
            raise NotImplementedError("dropout / stochastic depth are not implemented in the HIP tokenizer path")
        self.norm1 = norm_layer(dim)
        self.norm2 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)


class ViTEncoder(nn.Module):
    """Images -> (B, dim_tokens, H/P, W/P) latent features.  Same constructor as upstream."""

    def __init__(self, *, in_channels: int = 3, patch_size: int = 16, resolution: int = 256, dim_tokens: int = 768, depth: int = 12,
                 num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.0, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6), sincos_pos_emb: bool = True,
                 learnable_pos_emb: bool = False, patch_proj: bool = True, post_mlp: bool = False, ckpt_path: Optional[str] = None,
                 **ignore_kwargs):
        super().__init__()
        self.in_channels, self.dim_tokens, self.patch_proj = in_channels, dim_tokens, patch_proj
        self.P_H, self.P_W = pair(patch_size)
        self.H, self.W = pair(resolution)
        assert self.H % self.P_H == 0 and self.W % self.P_W == 0
        n_h, n_w = self.H // self.P_H, self.W // self.P_W
        if sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=n_h, w=n_w, embed_dim=dim_tokens), requires_grad=learnable_pos_emb)
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, dim_tokens, n_h, n_w))
            nn.init.trunc_normal_(self.pos_emb, std=0.02)
        # patch_proj=False (feature-map tokenizers: CLIP / DINOv2 / ImageBind): the input is already one vector per token, 1 x 1 projection
        k = (self.P_H, self.P_W) if patch_proj else (1, 1)
        self.proj = nn.Conv2d(in_channels, dim_tokens, kernel_size=k, stride=k)
        self.blocks = nn.Sequential(*[Block(dim=dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                                            attn_drop=attn_drop_rate, drop_path=drop_path_rate, norm_layer=norm_layer) for _ in range(depth)])
        if post_mlp:
            self.norm_mlp = norm_layer(dim_tokens)
            self.post_mlp = Mlp(dim_tokens, int(mlp_ratio * dim_tokens), act_layer=nn.Tanh)
        _init_vit_weights(self)
        nn.init.xavier_uniform_(self.proj.weight.data.view(self.proj.weight.shape[0], -1))

    def get_num_layers(self) -> int:
        return len(self.blocks)

    def forward(self, x):
        from fourm.vq.engine import encoder_forward
        return encoder_forward(self, x)


def _enc(dim, depth, heads):
    def build(in_channels, patch_size, resolution, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
              norm_layer=partial(nn.LayerNorm, eps=1e-6), sincos_pos_emb=True, learnable_pos_emb=False, patch_proj=True, post_mlp=False,
              ckpt_path=None):
        return ViTEncoder(in_channels=in_channels, patch_size=patch_size, resolution=resolution, dim_tokens=dim, depth=depth,
                          num_heads=heads, mlp_ratio=4, qkv_bias=True, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                          drop_path_rate=drop_path_rate, norm_layer=norm_layer, sincos_pos_emb=sincos_pos_emb,
                          learnable_pos_emb=learnable_pos_emb, patch_proj=patch_proj, post_mlp=post_mlp, ckpt_path=ckpt_path)
    return build


vit_s_enc, vit_b_enc, vit_l_enc = _enc(512, 8, 8), _enc(768, 12, 12), _enc(1024, 24, 16)


def _init_vit_weights(mod):
    """Upstream's initialisation (vit_models.py:584-612): xavier on every Linear with q / k / v treated separately, unit LayerNorms."""
    for name, m in mod.named_modules():
        if isinstance(m, nn.Linear):
            fused = 3 if "qkv" in name else 1
            if fused > 1:
                bound = math.sqrt(6. / float(m.weight.shape[0] // fused + m.weight.shape[1]))
                nn.init.uniform_(m.weight, -bound, bound)
            else:
                nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)


class ViTDecoder(nn.Module):
    """(B, dim_tokens, N_H, N_W) latent features -> (B, out_channels, H, W) images.  Same constructor as upstream (:528-546); the
    arithmetic runs in ``fourm.vq.engine`` (blocks on the trunk kernels, out_proj + patch re-assembly)."""

    def __init__(self, *, out_channels: int = 3, patch_size: int = 16, resolution: int = 256, dim_tokens: int = 768, depth: int = 12,
                 num_heads: int = 12, mlp_ratio: float = 4.0, qkv_bias: bool = True, drop_rate: float = 0.0, attn_drop_rate: float = 0.0,
                 drop_path_rate: float = 0.0, norm_layer: nn.Module = partial(nn.LayerNorm, eps=1e-6), sincos_pos_emb: bool = True,
                 learnable_pos_emb: bool = False, patch_proj: bool = True, post_mlp: bool = False, out_conv: bool = False, **ignore_kwargs):
        super().__init__()
        if out_conv:
            raise NotImplementedError("out_conv=True (ConvNeXt blocks behind the decoder) has no HIP path")
        self.out_channels, self.dim_tokens, self.patch_proj = out_channels, dim_tokens, patch_proj
        self.P_H, self.P_W = pair(patch_size)
        self.H, self.W = pair(resolution)
        assert self.H % self.P_H == 0 and self.W % self.P_W == 0, f"Image sizes {self.H}x{self.W} must be divisible by patch sizes {self.P_H}x{self.P_W}"
        n_h, n_w = self.H // self.P_H, self.W // self.P_W
        if sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=n_h, w=n_w, embed_dim=dim_tokens), requires_grad=learnable_pos_emb)
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, dim_tokens, n_h, n_w))
            nn.init.trunc_normal_(self.pos_emb, std=0.02)
        self.blocks = nn.Sequential(*[Block(dim=dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                                            attn_drop=attn_drop_rate, drop_path=drop_path_rate, norm_layer=norm_layer) for _ in range(depth)])
        if post_mlp:
            self.norm_mlp = norm_layer(dim_tokens)
            self.post_mlp = Mlp(dim_tokens, int(mlp_ratio * dim_tokens), act_layer=nn.Tanh)
        self.out_proj = nn.Linear(dim_tokens, out_channels * self.P_H * self.P_W if patch_proj else out_channels)
        _init_vit_weights(self)

    def get_num_layers(self) -> int:
        return len(self.blocks)

    def forward(self, x):
        raise RuntimeError("ViTDecoder computes inside VQVAE.decode_quant / forward (fourm.vq.engine); it has no stand-alone forward")


def _dec(dim, depth, heads):
    def build(out_channels, patch_size, resolution, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
              norm_layer=partial(nn.LayerNorm, eps=1e-6), sincos_pos_emb=True, learnable_pos_emb=False, patch_proj=True, post_mlp=False,
              out_conv=False):
        return ViTDecoder(out_channels=out_channels, patch_size=patch_size, resolution=resolution, dim_tokens=dim, depth=depth,
                          num_heads=heads, mlp_ratio=4, qkv_bias=True, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate,
                          drop_path_rate=drop_path_rate, norm_layer=norm_layer, sincos_pos_emb=sincos_pos_emb,
                          learnable_pos_emb=learnable_pos_emb, patch_proj=patch_proj, post_mlp=post_mlp, out_conv=out_conv)
    return build


vit_s_dec, vit_b_dec, vit_l_dec = _dec(512, 8, 8), _dec(768, 12, 12), _dec(1024, 24, 16)


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
