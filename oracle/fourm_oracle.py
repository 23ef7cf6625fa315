"""CPU oracle for the 4M masked encoder-decoder hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this module; the product path (``ml-4m_amd/``) never does.

This is an independent, functional (state-dict in, tensors out) restatement in plain fp32 PyTorch
of what the upstream model computes on the path

    embed -> concat -> select N / M tokens -> encoder -> context proj -> decoder -> logits / CE

following (file:line in /root/reference):
  * per-modality embedders ....... fourm/models/encoder_embeddings.py:87-121,184-211,280-309,387-421
                                   fourm/models/decoder_embeddings.py:98-152,226-268
  * concat + token selection ..... fourm/models/fm.py:245-438
  * decoder attention mask ....... fourm/models/fm.py:440-475
  * blocks ....................... fourm/models/fm_utils.py:93-219,310-366
  * model glue + losses .......... fourm/models/fm.py:477-691

Parity status: PINNED.  ``tests/golden/make_golden.py`` runs the unmodified upstream modules (imported
from /root/reference in the build container) on the same weights and inputs, checks this restatement
against them and writes the fixtures under ``tests/golden/*.npz`` that ``tests/test_oracle_golden.py``
replays without the reference tree.

Differences in *form* (not in results):
  * selection is an explicit stable partition (prefix sums), not a float argsort (fm.py:364-367);
  * the decoder modality order is an explicit argument (upstream draws it with ``random.sample``,
    fm.py:306);
  * ``emulate_bf16=True`` rounds to bfloat16 at exactly the places CUDA autocast does upstream
    (run_training_4m.py:723) so that the bf16 HIP path can be compared tightly.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# configuration records
# --------------------------------------------------------------------------------------------

def uint15_id(name: str) -> int:
    """sha256-derived 15-bit modality id (fourm/utils/misc.py:39-41)."""
    return int(hashlib.sha256(name.encode("utf-8")).hexdigest(), 16) % (1 << 15)


@dataclass
class ModSpec:
    """One modality as the trunk sees it.

    kind: 'tok'      image-like grid of discrete tokens (ImageToken*Embedding)
          'patch'    raw pixels, 16x16 patch projection (ImageEncoderEmbedding), encoder only
          'seq'      1-D token sequence (Sequence*Embedding)
          'seq_emb'  1-D sequence of dense embeddings (SequenceEmbEncoderEmbedding), encoder only
    """
    name: str
    kind: str
    vocab: int = 0
    n_pos: int = 0            # rows of pos_emb: grid cells for tok/patch, max_length for seq kinds
    patch: int = 16           # patch edge (kind == 'patch')
    channels: int = 3
    orig_dim: int = 0         # kind == 'seq_emb'
    padding_idx: Optional[int] = 0   # seq kinds: that embedding row gets no embedding-side gradient
    in_enc: bool = True
    in_dec: bool = True
    id: int = -2
    tensor_len: int = 0       # seq kinds: length of the padded id tensor the loader hands over

    def __post_init__(self):
        if self.id == -2:
            self.id = uint15_id(self.name)
        if self.tensor_len == 0:
            # loader convention: (max_tokens + 1) * 2 ids per sequence
            self.tensor_len = 2 * (self.n_pos + 1) if self.kind == "seq" else self.n_pos

    @property
    def is_seq(self) -> bool:
        return self.kind in ("seq", "seq_emb")

    @property
    def pos_rows(self) -> int:
        """Rows of the ``pos_emb`` entry.  Upstream slices the (1, 512, D) sin-cos table on its
        leading axis (encoder_embeddings.py:69, decoder_embeddings.py:74), so sequence modalities keep
        all 512 rows whatever their max_length."""
        return 512 if self.is_seq else self.n_pos


@dataclass
class TrunkCfg:
    dim: int = 768
    enc_depth: int = 12
    dec_depth: int = 12
    heads: int = 12
    mlp_ratio: float = 4.0
    gated: bool = True         # SwiGLU (fc1, fc3 -> fc2) vs plain MLP
    act: str = "silu"          # 'silu' | 'gelu'
    qkv_bias: bool = False
    proj_bias: bool = False
    mlp_bias: bool = False
    qk_norm: bool = False
    causal: bool = False       # decoder_causal_mask
    sep: bool = True           # decoder_sep_mask
    registers: int = 0
    zero_attn: bool = False    # allow_zero_attn of every Attention / CrossAttention (Block argument upstream)
    eps: float = 1e-6
    mods: List[ModSpec] = field(default_factory=list)

    @property
    def hidden(self) -> int:
        h = int(self.dim * self.mlp_ratio)
        return int(2 * h / 3) if self.gated else h

    def mod(self, name: str) -> ModSpec:
        for m in self.mods:
            if m.name == name:
                return m
        raise KeyError(name)


# --------------------------------------------------------------------------------------------
# numerics helpers
# --------------------------------------------------------------------------------------------

class _Num:
    """fp32 arithmetic, optionally rounding to bf16 where CUDA autocast would."""

    def __init__(self, emulate_bf16: bool):
        self.e = emulate_bf16

    def r(self, t: Tensor) -> Tensor:
        return t.to(torch.bfloat16).to(torch.float32) if self.e else t

    def linear(self, x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
        y = self.r(x) @ self.r(w).t()
        if b is not None:
            y = y + self.r(b)
        return self.r(y)

    def layer_norm(self, x: Tensor, w: Tensor, b: Optional[Tensor], eps: float) -> Tensor:
        mu = x.mean(-1, keepdim=True)
        var = ((x - mu) ** 2).mean(-1, keepdim=True)
        y = (x - mu) * torch.rsqrt(var + eps) * w
        return y + b if b is not None else y

    def act(self, x: Tensor, kind: str) -> Tensor:
        if kind == "silu":
            return self.r(x * torch.sigmoid(x))
        if kind == "gelu":
            return self.r(0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0)))))
        raise ValueError(kind)

    @property
    def neg_fill(self) -> float:
        return -torch.finfo(torch.bfloat16 if self.e else torch.float32).max


def sincos_1d(n: int, dim: int, temperature: float = 10000.0) -> Tensor:
    """(n, dim) table, sin half then cos half (fm_utils.py:32-44)."""
    half = dim // 2
    freq = 1.0 / (temperature ** (torch.arange(half, dtype=torch.float32) / half))
    ang = torch.arange(n, dtype=torch.float32)[:, None] * freq[None, :]
    return torch.cat([ang.sin(), ang.cos()], 1)


def sincos_2d(h: int, w: int, dim: int, temperature: float = 10000.0) -> Tensor:
    """(h*w, dim) table in the upstream convention (fm_utils.py:46-61): positions are enumerated
    with the *first* grid coordinate slow (meshgrid 'ij' over (w, h)), first half of the channels
    uses that slow coordinate."""
    q = dim // 4
    freq = 1.0 / (temperature ** (torch.arange(q, dtype=torch.float32) / q))
    a = torch.arange(w, dtype=torch.float32)[:, None].expand(w, h).reshape(-1)   # slow coordinate
    b = torch.arange(h, dtype=torch.float32)[None, :].expand(w, h).reshape(-1)   # fast coordinate
    aa, bb = a[:, None] * freq[None, :], b[:, None] * freq[None, :]
    return torch.cat([aa.sin(), aa.cos(), bb.sin(), bb.cos()], 1)


# --------------------------------------------------------------------------------------------
# stage 1: per-modality embedding of *all* positions + concat   (what upstream materialises)
# --------------------------------------------------------------------------------------------

def _seq_positions(mask: Tensor, limit: Optional[int] = None) -> Tensor:
    """Index of each unmasked element among the unmasked ones; 0 where masked (or >= limit)."""
    pos = (~mask).to(torch.int64).cumsum(1) - 1
    pos = pos.masked_fill(mask, 0)
    if limit is not None:
        pos = pos.masked_fill(pos >= limit, 0)
    return pos


def _patchify(img: Tensor, p: int) -> Tensor:
    """(B,C,H,W) -> (B, H/p*W/p, p*p*C) with feature order (row-in-patch, col-in-patch, channel)."""
    B, C, H, W = img.shape
    t = img.reshape(B, C, H // p, p, W // p, p).permute(0, 2, 4, 3, 5, 1)
    return t.reshape(B, (H // p) * (W // p), p * p * C)


def embed_encoder_modality(P: Dict[str, Tensor], ms: ModSpec, d: Dict[str, Tensor], num: _Num):
    pre = f"encoder_embeddings.{ms.name}."
    pos, modv = P[pre + "pos_emb"][0], P[pre + "mod_emb"][0]          # (L,D), (1,D)
    if ms.kind == "tok":
        ids = d["tensor"].reshape(d["tensor"].shape[0], -1).long()
        x = P[pre + "token_emb.weight"][ids]
        emb = (pos + modv)[None].expand(ids.shape[0], -1, -1)
    elif ms.kind == "patch":
        x = num.linear(_patchify(d["tensor"], ms.patch), P[pre + "proj.weight"], None)
        emb = (pos + modv)[None].expand(x.shape[0], -1, -1)
    elif ms.kind in ("seq", "seq_emb"):
        if ms.kind == "seq":
            x = P[pre + "token_emb.weight"][d["tensor"].long()]
        else:
            x = num.linear(d["tensor"], P[pre + "emb_proj.weight"], P[pre + "emb_proj.bias"])
        m = d["input_mask"].bool()
        pe = pos[_seq_positions(m)]
        pe = pe.masked_fill(m[..., None], 0.0)
        emb = pe + modv
    else:
        raise ValueError(ms.kind)
    return x, emb


def embed_decoder_modality(P: Dict[str, Tensor], ms: ModSpec, d: Dict[str, Tensor]):
    pre = f"decoder_embeddings.{ms.name}."
    pos, modv = P[pre + "pos_emb"][0], P[pre + "mod_emb"][0]
    if ms.kind == "tok":
        ids = d["tensor"].reshape(d["tensor"].shape[0], -1)
        x = P[pre + "token_emb.weight"][ids.long()]
        emb = (pos + modv)[None].expand(ids.shape[0], -1, -1)
    elif ms.kind == "seq":
        ids = d["tensor"]
        x = P[pre + "token_emb.weight"][ids.long()]
        m = d["target_mask"].bool()
        pe = pos[_seq_positions(m, limit=ms.n_pos)]
        pe = pe.masked_fill(m[..., None], 0.0)
        emb = pe + modv
    else:
        raise ValueError(ms.kind)
    return x, emb, ids


def stable_partition_keep(mask: Tensor, n_keep: int) -> Tensor:
    """Indices of the first ``n_keep`` elements after moving unmasked (False) positions to the front,
    order otherwise preserved.  Equals ``argsort(mask + arange*1e-6)[:, :n]`` of fm.py:364-367."""
    B, L = mask.shape
    valid = ~mask
    n_valid = valid.sum(1, keepdim=True)
    rank_valid = valid.to(torch.int64).cumsum(1) - 1
    rank_masked = n_valid + mask.to(torch.int64).cumsum(1) - 1
    dest = torch.where(valid, rank_valid, rank_masked)                # a permutation per row
    order = torch.empty_like(dest)
    order.scatter_(1, dest, torch.arange(L).expand(B, L))
    return order[:, :n_keep]


def _gather_rows(t: Tensor, idx: Tensor) -> Tensor:
    if t.dim() == 3:
        return torch.gather(t, 1, idx[..., None].expand(-1, -1, t.shape[2]))
    return torch.gather(t, 1, idx)


def select_encoder(P, cfg: TrunkCfg, mod_dict, n_keep: int, num: _Num) -> Dict[str, Tensor]:
    xs, es, ms_, ids_ = [], [], [], []
    for name, d in mod_dict.items():
        try:
            spec = cfg.mod(name)
        except KeyError:
            continue
        if not spec.in_enc:
            continue
        x, e = embed_encoder_modality(P, spec, d, num)
        m = d["input_mask"].bool().reshape(x.shape[0], -1)
        xs.append(x.float()); es.append(e); ms_.append(m)
        ids_.append(torch.full(m.shape, spec.id, dtype=torch.int16))
    x_all, e_all = torch.cat(xs, 1), torch.cat(es, 1)
    m_all, id_all = torch.cat(ms_, 1), torch.cat(ids_, 1)
    keep = stable_partition_keep(m_all, n_keep)
    tok, emb = _gather_rows(x_all, keep), _gather_rows(e_all, keep)
    msk, mod = _gather_rows(m_all, keep), _gather_rows(id_all, keep)
    B = tok.shape[0]
    if cfg.registers > 0:
        reg = P["register_tokens"].expand(B, -1, -1)
        tok = torch.cat([reg, tok], 1)
        emb = torch.cat([torch.zeros_like(reg), emb], 1)
        msk = torch.cat([torch.zeros(B, cfg.registers, dtype=torch.bool), msk], 1)
        mod = torch.cat([torch.full((B, cfg.registers), -1, dtype=torch.int16), mod], 1)
    tok = tok.masked_fill(msk[..., None], 0.0)
    emb = emb.masked_fill(msk[..., None], 0.0)
    mod = mod.masked_fill(msk, -1)
    return dict(tokens=tok, emb=emb, mask=msk[:, None, :], mod_mask=mod, ids_keep=keep)


def decoder_attention_mask(dam: Tensor, mod: Tensor, causal: bool, sep: bool) -> Tensor:
    """True = blocked.  Row n1 sees the columns below cumsum(dam)[n1] (fm.py:459-468) of its own
    modality only (fm.py:470-473)."""
    B, M = dam.shape
    col = torch.arange(M)
    if causal:
        blocked = (col[None, :] > col[:, None])[None].expand(B, M, M)
    else:
        blocked = col[None, None, :] >= dam.cumsum(-1)[:, :, None]
    if sep:
        blocked = blocked | (mod[:, :, None] != mod[:, None, :])
    return blocked


def select_decoder(P, cfg: TrunkCfg, mod_dict, n_keep: int, order: Sequence[str]) -> Dict[str, Tensor]:
    xs, es, ms_, tg, am, ids_ = [], [], [], [], [], []
    for name in order:
        d, spec = mod_dict[name], cfg.mod(name)
        x, e, ids = embed_decoder_modality(P, spec, d)
        tm = d["target_mask"].bool().reshape(x.shape[0], -1)
        dam = d["decoder_attention_mask"].reshape(x.shape[0], -1)
        if spec.is_seq:
            # teacher forcing: input = token t, target = token t+1 (fm.py:309-319)
            xs.append(x[:, :-1]); es.append(e[:, :-1]); tg.append(ids[:, 1:])
            ms_.append(tm[:, 1:] | tm[:, :-1]); am.append(dam[:, :-1])
            L = x.shape[1] - 1
        else:
            # grid modalities are queried with the mask token (fm.py:322)
            xs.append(P["mask_token"].expand(x.shape[0], x.shape[1], -1)); es.append(e)
            tg.append(ids); ms_.append(tm); am.append(dam)
            L = x.shape[1]
        ids_.append(torch.full((x.shape[0], L), spec.id, dtype=torch.int16))
    x_all, e_all = torch.cat(xs, 1), torch.cat(es, 1)
    m_all, t_all = torch.cat(ms_, 1), torch.cat(tg, 1)
    a_all, id_all = torch.cat(am, 1), torch.cat(ids_, 1)
    keep = stable_partition_keep(m_all, n_keep)
    tok, emb = _gather_rows(x_all, keep), _gather_rows(e_all, keep)
    msk, tgt = _gather_rows(m_all, keep), _gather_rows(t_all, keep)
    dam, mod = _gather_rows(a_all, keep), _gather_rows(id_all, keep)
    tok = tok.masked_fill(msk[..., None], 0.0)
    emb = emb.masked_fill(msk[..., None], 0.0)
    tgt = tgt.masked_fill(msk, 0)
    attn_mask = decoder_attention_mask(dam, mod, cfg.causal, cfg.sep)   # before pads lose their id
    mod = mod.masked_fill(msk, -1)
    return dict(tokens=tok, emb=emb, mask=msk[:, None, :], target_ids=tgt.long(), attn_mask=attn_mask,
                mod_mask=mod, ids_keep=keep, dam=dam)


# --------------------------------------------------------------------------------------------
# stage 2: transformer trunk
# --------------------------------------------------------------------------------------------

def _ln(P, pre: str, x: Tensor, cfg: TrunkCfg, num: _Num) -> Tensor:
    return num.layer_norm(x, P[pre + ".weight"], P.get(pre + ".bias"), cfg.eps)


def _softmax_masked(scores: Tensor, blocked: Optional[Tensor], num: _Num, zero_attn: bool = False) -> Tensor:
    if blocked is not None:
        scores = scores.masked_fill(blocked, num.neg_fill)
    if zero_attn:            # softmax1 (fm_utils.py:28-30, :171-172): a zero logit joins the softmax, its probability is dropped
        return torch.softmax(F.pad(scores, (0, 1)), -1)[..., :-1]
    return torch.softmax(scores, -1)


def _heads(t: Tensor, H: int) -> Tensor:
    B, L, C = t.shape
    return t.reshape(B, L, H, C // H).transpose(1, 2)


def _attend(q, k, v, blocked, cfg: TrunkCfg, num: _Num, P=None, pre=None) -> Tensor:
    H = cfg.heads
    q, k, v = _heads(q, H), _heads(k, H), _heads(v, H)
    if cfg.qk_norm:
        q = num.layer_norm(q, P[pre + ".q_norm.weight"], P.get(pre + ".q_norm.bias"), cfg.eps)
        k = num.layer_norm(k, P[pre + ".k_norm.weight"], P.get(pre + ".k_norm.bias"), cfg.eps)
    s = num.r(num.r(num.r(q) @ num.r(k).transpose(-1, -2)) * (q.shape[-1] ** -0.5))
    p = _softmax_masked(s, blocked, num, cfg.zero_attn)
    o = num.r(num.r(p) @ num.r(v))
    return o.transpose(1, 2).reshape(o.shape[0], o.shape[2], -1)


def self_attention(P, pre, x, blocked, cfg, num):
    qkv = num.linear(x, P[pre + ".qkv.weight"], P.get(pre + ".qkv.bias"))
    q, k, v = qkv.chunk(3, -1)
    o = _attend(q, k, v, None if blocked is None else blocked[:, None], cfg, num, P, pre)
    return num.linear(o, P[pre + ".proj.weight"], P.get(pre + ".proj.bias"))


def cross_attention(P, pre, x, ctx, blocked, cfg, num):
    q = num.linear(x, P[pre + ".q.weight"], P.get(pre + ".q.bias"))
    kv = num.linear(ctx, P[pre + ".kv.weight"], P.get(pre + ".kv.bias"))
    k, v = kv.chunk(2, -1)
    o = _attend(q, k, v, None if blocked is None else blocked[:, None], cfg, num, P, pre)
    return num.linear(o, P[pre + ".proj.weight"], P.get(pre + ".proj.bias"))


def mlp(P, pre, x, cfg, num):
    if cfg.gated:
        g = num.act(num.linear(x, P[pre + ".fc1.weight"], P.get(pre + ".fc1.bias")), cfg.act)
        u = num.linear(x, P[pre + ".fc3.weight"], P.get(pre + ".fc3.bias"))
        return num.linear(num.r(g * u), P[pre + ".fc2.weight"], P.get(pre + ".fc2.bias"))
    h = num.act(num.linear(x, P[pre + ".fc1.weight"], P.get(pre + ".fc1.bias")), cfg.act)
    return num.linear(h, P[pre + ".fc2.weight"], P.get(pre + ".fc2.bias"))


def drop_path_scale(u: Tensor, drop_prob: float) -> Tensor:
    """DropPath (fm_utils.py:64-76) with its uniform draw explicit: floor(keep_prob + u) / keep_prob per sample, shaped (B, 1, 1)."""
    keep = 1.0 - drop_prob
    return (torch.floor(keep + u.float()) / keep).reshape(-1, 1, 1)


def _dp(drop, side, i, j):
    """Scale of branch j of block i (``drop`` = {"enc": [[s_attn, s_mlp], ...], "dec": [[s_self, s_cross, s_mlp], ...]}), or 1."""
    return 1.0 if drop is None else drop[side][i][j]


def encoder_forward(P, cfg, x, enc_mask, num, taps=None, drop=None):
    for i in range(cfg.enc_depth):
        pre = f"encoder.{i}"
        x = x + _dp(drop, "enc", i, 0) * self_attention(P, pre + ".attn", _ln(P, pre + ".norm1", x, cfg, num), enc_mask, cfg, num)
        x = x + _dp(drop, "enc", i, 1) * mlp(P, pre + ".mlp", _ln(P, pre + ".norm2", x, cfg, num), cfg, num)
        if taps is not None:
            taps[f"enc_block{i}"] = x
    return _ln(P, "encoder_norm", x, cfg, num)


def decoder_forward(P, cfg, y, ctx, enc_mask, sa_blocked, num, taps=None, drop=None):
    for i in range(cfg.dec_depth):
        pre = f"decoder.{i}"
        y = y + _dp(drop, "dec", i, 0) * self_attention(P, pre + ".self_attn", _ln(P, pre + ".norm1", y, cfg, num), sa_blocked, cfg, num)
        y = y + _dp(drop, "dec", i, 1) * cross_attention(P, pre + ".cross_attn", _ln(P, pre + ".query_norm", y, cfg, num),
                                                        _ln(P, pre + ".context_norm", ctx, cfg, num), enc_mask, cfg, num)
        y = y + _dp(drop, "dec", i, 2) * mlp(P, pre + ".mlp", _ln(P, pre + ".norm2", y, cfg, num), cfg, num)
        if taps is not None:
            taps[f"dec_block{i}"] = y
    return _ln(P, "decoder_norm", y, cfg, num)


# --------------------------------------------------------------------------------------------
# stage 3: heads and losses
# --------------------------------------------------------------------------------------------

def modality_losses(P, cfg, y, target_ids, dec_mod_mask, dec_names, loss_type, num):
    per_mod, counts = {}, {}
    for name in dec_names:
        spec = cfg.mod(name)
        sel = dec_mod_mask == spec.id
        rows = y[sel]
        W = P[f"decoder_embeddings.{name}.to_logits.weight"]
        if rows.shape[0] == 0:
            per_mod[name] = torch.zeros(1)
            counts[name] = 0
            continue
        logits = num.linear(rows, W, None)
        per_mod[name] = F.cross_entropy(logits, target_ids[sel], reduction="mean")
        counts[name] = logits.numel()
    if loss_type in ("mod", "modality"):
        total = sum(per_mod.values()) / len(per_mod)
    elif loss_type == "token":
        total = sum(per_mod[n] * counts[n] for n in per_mod) / sum(counts.values())
    else:
        raise ValueError("Invalid loss type")
    return total, per_mod


def fourm_forward(P: Dict[str, Tensor], cfg: TrunkCfg, mod_dict: Dict[str, Dict[str, Tensor]],
                  num_encoder_tokens: int, num_decoder_tokens: int, dec_order: Sequence[str],
                  loss_type: str = "mod", return_logits: bool = False, emulate_bf16: bool = False,
                  taps: Optional[dict] = None, drop: Optional[dict] = None):
    """Whole-model forward (fm.py:640-691).  ``dec_order`` lists the decoder modalities present in
    ``mod_dict`` in the order they are concatenated.  Returns (loss, {mod: loss}) or {mod: logits}.
    ``taps`` (a dict) receives intermediate tensors at named cut points."""
    num = _Num(emulate_bf16)
    enc = select_encoder(P, cfg, mod_dict, num_encoder_tokens, num)
    dec = select_decoder(P, cfg, mod_dict, num_decoder_tokens, dec_order)
    if taps is not None:
        taps.update({"enc_" + k: v for k, v in enc.items()})
        taps.update({"dec_" + k: v for k, v in dec.items()})
    x = encoder_forward(P, cfg, enc["tokens"] + enc["emb"], enc["mask"], num, taps, drop)      # drop: DropPath scales in training mode
    ctx = num.linear(x, P["decoder_proj_context.weight"], P["decoder_proj_context.bias"]) + enc["emb"]
    y = decoder_forward(P, cfg, dec["tokens"] + dec["emb"], ctx, enc["mask"], dec["attn_mask"], num, taps, drop)
    if taps is not None:
        taps["enc_out"], taps["context"], taps["dec_out"] = x, ctx, y
    dec_names = [n for n in mod_dict if n in dec_order]     # heads iterate in mod_dict order
    if return_logits:
        return {n: num.linear(y, P[f"decoder_embeddings.{n}.to_logits.weight"], None) for n in dec_names}
    return modality_losses(P, cfg, y, dec["target_ids"], dec["mod_mask"], dec_names, loss_type, num)


# --------------------------------------------------------------------------------------------
# parameter construction (reference state_dict layout)
# --------------------------------------------------------------------------------------------

def param_shapes(cfg: TrunkCfg, share_embedding: bool = True) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, kind) for every state_dict entry.  kind in
    {'linear','qkv','kv','norm_w','norm_b','bias','emb','tok','posbuf','buf0'}; 'posbuf' entries are
    the fixed sin-cos tables, 'buf0' the zero bias buffers of the bias-free LayerNorm."""
    D, Hd = cfg.dim, cfg.hidden
    out: Dict[str, Tuple[Tuple[int, ...], str]] = {"mask_token": ((1, 1, D), "tok")}
    if cfg.registers:
        out["register_tokens"] = ((1, cfg.registers, D), "tok")

    def norm(pre, width=D):
        out[pre + ".weight"] = ((width,), "norm_w")
        out[pre + ".bias"] = ((width,), "norm_b")

    def lin(pre, o, i, bias, kind="linear"):
        out[pre + ".weight"] = ((o, i), kind)
        if bias:
            out[pre + ".bias"] = ((o,), "bias")

    def mlp_(pre):
        lin(pre + ".fc1", Hd, D, cfg.mlp_bias)
        lin(pre + ".fc2", D, Hd, cfg.mlp_bias)
        if cfg.gated:
            lin(pre + ".fc3", Hd, D, cfg.mlp_bias)

    def qkn(pre):
        if cfg.qk_norm:
            norm(pre + ".q_norm", D // cfg.heads)
            norm(pre + ".k_norm", D // cfg.heads)

    for m in cfg.mods:
        for side, present in (("encoder", m.in_enc), ("decoder", m.in_dec)):
            if not present:
                continue
            pre = f"{side}_embeddings.{m.name}"
            out[pre + ".mod_emb"] = ((1, 1, D), "tok")
            out[pre + ".pos_emb"] = ((1, m.pos_rows, D), "pos")
            if m.kind in ("tok", "seq"):
                out[pre + ".token_emb.weight"] = ((m.vocab, D), "emb")
            if side == "encoder" and m.kind == "patch":
                out[pre + ".proj.weight"] = ((D, m.patch * m.patch * m.channels), "linear")
            if side == "encoder" and m.kind == "seq_emb":
                lin(pre + ".emb_proj", D, m.orig_dim, True)
            if side == "decoder":
                out[pre + ".to_logits.weight"] = ((m.vocab, D), "emb" if share_embedding else "linear")
    for i in range(cfg.enc_depth):
        pre = f"encoder.{i}"
        norm(pre + ".norm1"); norm(pre + ".norm2")
        lin(pre + ".attn.qkv", 3 * D, D, cfg.qkv_bias, "qkv"); lin(pre + ".attn.proj", D, D, cfg.proj_bias)
        qkn(pre + ".attn"); mlp_(pre + ".mlp")
    norm("encoder_norm")
    lin("decoder_proj_context", D, D, True)
    for i in range(cfg.dec_depth):
        pre = f"decoder.{i}"
        for n in ("norm1", "query_norm", "context_norm", "norm2"):
            norm(f"{pre}.{n}")
        lin(pre + ".self_attn.qkv", 3 * D, D, cfg.qkv_bias, "qkv"); lin(pre + ".self_attn.proj", D, D, cfg.proj_bias)
        qkn(pre + ".self_attn")
        lin(pre + ".cross_attn.q", D, D, cfg.qkv_bias); lin(pre + ".cross_attn.kv", 2 * D, D, cfg.qkv_bias, "kv")
        lin(pre + ".cross_attn.proj", D, D, cfg.proj_bias); qkn(pre + ".cross_attn")
        mlp_(pre + ".mlp")
    norm("decoder_norm")
    return out


def seeded_tensor(name: str, shape: Sequence[int], scale: float, seed: int = 0) -> Tensor:
    """Deterministic N(0, scale^2) tensor that depends only on (name, shape, seed): lets two processes
    (the upstream model here, the HIP model on the GPU box) agree on weights without shipping them."""
    g = torch.Generator(device="cpu")
    g.manual_seed((int(hashlib.sha256(name.encode()).hexdigest(), 16) + seed) % (2 ** 63 - 1))
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32) * scale


def seeded_state_dict(cfg: TrunkCfg, seed: int = 0, share_embedding: bool = True,
                      learned_pos: Sequence[str] = (), norm_bias: bool = False) -> Dict[str, Tensor]:
    """A full reference-layout state_dict with weights drawn by ``seeded_tensor``.

    Scales are chosen so activations stay O(1) through the stack (the values are test data, not an
    initialisation scheme).  Fixed position tables are rebuilt with the upstream formulas."""
    sd: Dict[str, Tensor] = {}
    shapes = param_shapes(cfg, share_embedding)
    for name, (shape, kind) in shapes.items():
        if kind == "pos":
            modname = name.split(".")[1]
            ms = cfg.mod(modname)
            if modname in learned_pos:
                sd[name] = seeded_tensor(name, shape, 0.02, seed)
            elif ms.is_seq:
                sd[name] = sincos_1d(512, cfg.dim)[None].clone()
            else:
                side = int(round(math.sqrt(ms.n_pos)))
                sd[name] = sincos_2d(side, side, cfg.dim)[None].clone()
        elif kind == "norm_w":
            sd[name] = 1.0 + seeded_tensor(name, shape, 0.1, seed)
        elif kind == "norm_b":
            # bias-free LayerNorm keeps an all-zero ``bias`` *buffer* in the state_dict (fm_utils.py:99-102)
            sd[name] = seeded_tensor(name, shape, 0.05, seed) if norm_bias else torch.zeros(shape)
        elif kind == "bias":
            sd[name] = seeded_tensor(name, shape, 0.02, seed)
        elif kind in ("emb", "tok"):
            sd[name] = seeded_tensor(name, shape, 0.02 if kind == "emb" else 0.05, seed)
        else:  # linear / qkv / kv
            sd[name] = seeded_tensor(name, shape, 1.0 / math.sqrt(shape[1]), seed)
    # sharing: encoder/decoder modality embedding (fm.py:176-180) and tied heads
    # (decoder_embeddings.py:89-91)
    for m in cfg.mods:
        if m.in_enc and m.in_dec:
            sd[f"decoder_embeddings.{m.name}.mod_emb"] = sd[f"encoder_embeddings.{m.name}.mod_emb"]
        if m.in_dec and share_embedding:
            sd[f"decoder_embeddings.{m.name}.to_logits.weight"] = sd[f"decoder_embeddings.{m.name}.token_emb.weight"]
    return sd


# --------------------------------------------------------------------------------------------
# named configurations (fm.py:939-1031) and the mod-7 modality set
# --------------------------------------------------------------------------------------------

def mod7_specs(image: int = 224, patch: int = 16) -> List[ModSpec]:
    g = (image // patch) ** 2
    tok = lambda n, v: ModSpec(f"{n}@{image}", "tok", vocab=v, n_pos=g)
    return [
        ModSpec("caption", "seq", vocab=30000, n_pos=256),
        ModSpec("det", "seq", vocab=30000, n_pos=256),
        ModSpec(f"rgb@{image}", "patch", n_pos=g, patch=patch, in_dec=False),
        tok("tok_clip", 8192), tok("tok_depth", 8192), tok("tok_normal", 8192),
        tok("tok_rgb", 16384), tok("tok_semseg", 4096),
    ]


def mod21_specs() -> List[ModSpec]:
    """The 19 input / 17 target modalities of the 4M-21 mixture (cfgs/default/4m/data/cc12m+coyo+c4/main/
    mix_mod21_all2allmix_rgb2all_capT5bias_C4.yaml:7-8) with the registry's vocabularies and lengths
    (fourm/data/modality_info.py:32-383): ViT-B/14 feature tokenizers give 16 x 16 grids, the two global-feature modalities 16
    tokens with a learned position table (patch 56 over 224)."""
    tok = lambda n, v, p=16: ModSpec(n, "tok", vocab=v, n_pos=(224 // p) ** 2, patch=p)
    seq = lambda n, L: ModSpec(n, "seq", vocab=30000, n_pos=L)
    return [
        seq("caption", 256), ModSpec("t5_caption", "seq_emb", n_pos=77, orig_dim=4096, in_dec=False), seq("det", 256), seq("metadata", 40),
        ModSpec("rgb@224", "patch", n_pos=196, patch=16, in_dec=False),
        tok("tok_rgb@224", 16384), tok("tok_normal@224", 8192), tok("tok_depth@224", 8192), tok("tok_semseg@224", 4096),
        tok("tok_clip@224", 8192), ModSpec("human_poses", "seq", vocab=30000, n_pos=263, tensor_len=2 * (275 + 1)),   # max_tokens 275
        tok("tok_dinov2@224", 8192, 14), tok("tok_dinov2_global", 8192, 56),
        tok("tok_imagebind@224", 8192, 14), tok("tok_imagebind_global", 8192, 56), tok("tok_sam_edge@224", 8192),
        tok("tok_canny_edge@224", 8192), seq("color_palette", 23), seq("sam_instance", 290),
    ]


MOD21_LEARNED_POS = ("tok_dinov2_global", "tok_imagebind_global")


def named_cfg(size: str, mods: List[ModSpec]) -> TrunkCfg:
    table = {"tiny": (384, 6, 6), "small": (512, 8, 8), "base": (768, 12, 12), "large": (1024, 24, 16),
             "xlarge": (2048, 24, 32)}
    dim, depth, heads = table[size]
    return TrunkCfg(dim=dim, enc_depth=depth, dec_depth=depth, heads=heads, gated=True, act="silu", mods=mods)


# --------------------------------------------------------------------------------------------
# synthetic batches
# --------------------------------------------------------------------------------------------

def synthetic_mod_dict(cfg: TrunkCfg, batch: int, n_in: int, n_out: int, seed: int = 0,
                       no_target: Sequence[str] = ()) -> Dict[str, Dict[str, Tensor]]:
    """Random-token multimodal batch with per-sample token budgets split over the modalities.

    Sample b draws its budgets from ``numpy.random.RandomState(1000 + seed + b)``: a multinomial over
    the encoder (resp. decoder) modalities, clipped to what each modality can hold.  Grid modalities
    take a random permutation of their cells (inputs first, then targets, so the two never overlap);
    sequences put inputs at [0,k_in) and targets at [k_in, k_in+k_out).  Modalities listed in
    ``no_target`` never receive target tokens (exercises the empty-head branch, fm.py:593-595)."""
    import numpy as np
    enc = [m for m in cfg.mods if m.in_enc]
    dec = [m for m in cfg.mods if m.in_dec and m.name not in no_target]
    out: Dict[str, Dict[str, Tensor]] = {}
    g = torch.Generator().manual_seed(seed)
    for m in sorted(cfg.mods, key=lambda s: s.name):
        if m.kind == "patch":
            side = int(round(math.sqrt(m.n_pos))) * m.patch
            t = torch.randn(batch, m.channels, side, side, generator=g)
            L = m.n_pos
        elif m.kind == "tok":
            side = int(round(math.sqrt(m.n_pos)))
            t = torch.randint(0, m.vocab, (batch, side, side), generator=g, dtype=torch.int64)
            L = m.n_pos
        elif m.kind == "seq":
            L = m.tensor_len
            t = torch.randint(5, m.vocab, (batch, L), generator=g, dtype=torch.int64).to(torch.int32)
        else:
            t = torch.randn(batch, m.n_pos, m.orig_dim, generator=g)
            L = m.n_pos
        out[m.name] = dict(tensor=t, input_mask=torch.ones(batch, L, dtype=torch.bool),
                           target_mask=torch.ones(batch, L, dtype=torch.bool),
                           decoder_attention_mask=torch.zeros(batch, L, dtype=torch.int32))
    for b in range(batch):
        rs = np.random.RandomState(1000 + seed + b)
        room = lambda m: m.n_pos if not m.is_seq else min(m.n_pos, m.tensor_len // 2)
        cap_in = {m.name: room(m) for m in enc}
        k_in = _budget(rs, [cap_in[m.name] for m in enc], n_in)
        used = dict(zip([m.name for m in enc], k_in))
        cap_out = [room(m) - (used.get(m.name, 0) if not m.is_seq else 0) for m in dec]
        k_out = dict(zip([m.name for m in dec], _budget(rs, cap_out, n_out)))
        for m in cfg.mods:
            d = out[m.name]
            ki, ko = used.get(m.name, 0), k_out.get(m.name, 0)
            if m.is_seq:
                d["input_mask"][b, :ki] = False
                if ko:
                    d["target_mask"][b, ki:ki + ko] = False
                    d["decoder_attention_mask"][b, ki:ki + ko] = 1
            else:
                perm = torch.from_numpy(rs.permutation(m.n_pos))
                d["input_mask"][b, perm[:ki]] = False
                if ko:
                    tgt = perm[ki:ki + ko]
                    d["target_mask"][b, tgt] = False
                    d["decoder_attention_mask"][b, int(tgt.min())] = ko
    return out


def _budget(rs, caps: List[int], total: int) -> List[int]:
    import numpy as np
    caps = np.asarray(caps, dtype=np.int64)
    k = rs.multinomial(total, np.ones(len(caps)) / len(caps))
    k = np.minimum(k, caps)
    # hand the clipped remainder to modalities with room, in order
    rest = total - int(k.sum())
    for i in range(len(caps)):
        if rest <= 0:
            break
        add = min(rest, int(caps[i] - k[i]))
        k[i] += add
        rest -= add
    return [int(v) for v in k]


# --------------------------------------------------------------------------------------------
# data side: input / target masks of an image-like modality
# --------------------------------------------------------------------------------------------
def image_mask(noise, input_budget: int, target_budget):
    """``UnifiedMasking.image_mask`` (fourm/data/masking.py:237-266) with the random vector made explicit (upstream draws
    ``noise = torch.rand(num_tokens)`` itself).  Returns (input_mask bool (L), target_mask bool (L), decoder_attention_mask int32 (L)).

        ids_shuffle = argsort(noise);  input_mask[i] = ids_shuffle[i] >= input_budget   (a gather of [0]*k_in + [1]*rest: position i is
        an INPUT iff the index of the i-th smallest noise value is below the budget - upstream's gather direction, kept as is);
        target_mask[i] = not (input_budget <= ids_shuffle[i] < input_budget + target_budget)   (or ~input_mask when target_budget is None);
        decoder_attention_mask = 0 except at the first target position, which carries the number of targets."""
    noise = torch.as_tensor(noise, dtype=torch.float32)
    L_ = noise.shape[0]
    ids = torch.argsort(noise, dim=0)
    input_mask = ids >= int(input_budget)
    if target_budget is None:
        target_mask = ~input_mask
    else:
        target_mask = ~((ids >= int(input_budget)) & (ids < int(input_budget) + int(target_budget)))
    dam = torch.zeros(L_, dtype=torch.int32)
    first = int(torch.argmin(target_mask.float() + torch.arange(L_) * 1e-6))
    dam[first] = int((~target_mask).sum())
    return input_mask, target_mask, dam
