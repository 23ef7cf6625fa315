"""Encoder-side modality embedders: parameter owners + the descriptors the fused
select-and-embed kernel consumes.

Constructor signatures, attribute and parameter names follow upstream
``fourm/models/encoder_embeddings.py`` (classes at :22, :123, :214, :312) so that checkpoints and the
``MODALITY_INFO`` registry work unchanged.  Upstream's ``forward`` embeds *every* position of the
modality; here ``FourM.forward_mask_encoder`` embeds only the positions that survive selection, in one
kernel over all modalities (csrc/select_embed.hip).
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .fm_utils import build_1d_sincos_posemb, build_2d_sincos_posemb, pair

KIND_TOK, KIND_PATCH, KIND_SEQ, KIND_SEQ_EMB = 0, 1, 2, 3


class _Embedder(nn.Module):
    """Shared machinery: positional table (fixed sin-cos buffer or learned parameter), modality
    embedding, weight-decay policy."""
    kind = -1

    def _make_pos(self, table: torch.Tensor, learned_rows: int, init_std: float):
        if self.sincos_pos_emb:
            self.register_buffer("pos_emb", table)
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, learned_rows, self.dim_tokens))
            nn.init.normal_(self.pos_emb, std=init_std)
        self.mod_emb = nn.Parameter(torch.zeros(1, 1, self.dim_tokens))
        nn.init.normal_(self.mod_emb, std=init_std)

    @torch.jit.ignore
    def no_weight_decay(self):
        return set()

    def forward(self, d):
        """Upstream contract: adds ``x`` (token / projected rows) and ``emb`` (position + modality embedding) for
        every position to ``d`` (fp32, (B, L, D)).  Inference only; training embeds inside FourM.forward."""
        from fourm.hip import functional
        d["x"], d["emb"] = functional.embed_modality(self, d, is_dec=False)
        return d


class _SeqPos(_Embedder):
    def _seq_pos(self, init_std):
        if self.sincos_pos_emb and self.max_length > self.max_sincos_pos_emb:
            raise ValueError(f"Max length ({self.max_length}) is greater than the number of posembs ({self.max_sincos_pos_emb}")
        # NB upstream keeps all max_sincos_pos_emb rows: its [:max_length] slices the leading axis of a
        # (1, 512, D) tensor (encoder_embeddings.py:69) — the state_dict shape is (1, 512, D).
        table = build_1d_sincos_posemb(max_len=self.max_sincos_pos_emb, embed_dim=self.dim_tokens)[:self.max_length]
        self._make_pos(table, self.max_length, init_std)


class SequenceEncoderEmbedding(_SeqPos):
    """Token sequences (captions, detection strings, ...).   [upstream :22-121]"""
    kind = KIND_SEQ

    def __init__(self, vocab_size: int, max_length: int, dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True,
                 max_sincos_pos_emb: int = 512, padding_idx: int = 0):
        super().__init__()
        self.vocab_size, self.max_length, self.dim_tokens = vocab_size, max_length, dim_tokens
        self.sincos_pos_emb, self.padding_idx, self.max_sincos_pos_emb = sincos_pos_emb, padding_idx, max_sincos_pos_emb
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768, init_std=0.02):
        self.dim_tokens = dim_tokens
        self._seq_pos(init_std)
        self.token_emb = nn.Embedding(self.vocab_size, self.dim_tokens, padding_idx=self.padding_idx)


class ImageTokenEncoderEmbedding(_Embedder):
    """Grids of discrete tokens (tokenized RGB / depth / ...).   [upstream :123-211]"""
    kind = KIND_TOK

    def __init__(self, vocab_size: int, patch_size: Union[int, Tuple[int, int]] = 16, dim_tokens: Optional[int] = None,
                 sincos_pos_emb: bool = True, image_size: Union[int, Tuple[int]] = 224, **kwargs):
        super().__init__()
        self.vocab_size, self.patch_size, self.dim_tokens = vocab_size, pair(patch_size), dim_tokens
        self.sincos_pos_emb, self.image_size = sincos_pos_emb, pair(image_size)
        self.num_patches = (self.image_size[0] // self.patch_size[0]) * (self.image_size[1] // self.patch_size[1])
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def _grid(self):
        return self.image_size[0] // self.patch_size[0], self.image_size[1] // self.patch_size[1]

    def init(self, dim_tokens: int = 768, init_std=0.02):
        self.dim_tokens = dim_tokens
        h, w = self._grid()
        self._make_pos(build_2d_sincos_posemb(h=h, w=w, embed_dim=dim_tokens) if self.sincos_pos_emb else None, h * w, init_std)
        self.token_emb = nn.Embedding(self.vocab_size, self.dim_tokens)


class ImageEncoderEmbedding(ImageTokenEncoderEmbedding):
    """Raw pixels: (ph pw c)-ordered patches through a bias-free Linear.   [upstream :214-309]"""
    kind = KIND_PATCH

    def __init__(self, num_channels: int, patch_size: Union[int, Tuple[int, int]], dim_tokens: Optional[int] = None,
                 sincos_pos_emb: bool = True, image_size: Union[int, Tuple[int]] = 224):
        self.num_channels = num_channels
        super().__init__(vocab_size=0, patch_size=patch_size, dim_tokens=dim_tokens, sincos_pos_emb=sincos_pos_emb,
                         image_size=image_size)
        del self.vocab_size

    def init(self, dim_tokens: int = 768, init_std=0.02):
        self.dim_tokens = dim_tokens
        h, w = self._grid()
        self._make_pos(build_2d_sincos_posemb(h=h, w=w, embed_dim=dim_tokens) if self.sincos_pos_emb else None, h * w, init_std)
        self.proj = nn.Linear(self.num_channels * self.patch_size[0] * self.patch_size[1], dim_tokens, bias=False)


class SequenceEmbEncoderEmbedding(_SeqPos):
    """Sequences of dense embeddings (T5-XXL caption features).   [upstream :312-421]"""
    kind = KIND_SEQ_EMB

    def __init__(self, max_length: int, dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, max_sincos_pos_emb: int = 512,
                 padding_idx: int = 0, orig_emb_dim: int = 4096, bottleneck_dim: int = 64, use_bottleneck: bool = False):
        super().__init__()
        if use_bottleneck:
            raise NotImplementedError("use_bottleneck=True has no HIP path (no upstream config enables it)")
        self.max_length, self.dim_tokens, self.sincos_pos_emb = max_length, dim_tokens, sincos_pos_emb
        self.padding_idx, self.max_sincos_pos_emb, self.orig_emb_dim = padding_idx, max_sincos_pos_emb, orig_emb_dim
        self.use_bottleneck = use_bottleneck
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768, init_std=0.02):
        self.dim_tokens = dim_tokens
        self._seq_pos(init_std)
        self.emb_proj = nn.Linear(self.orig_emb_dim, dim_tokens)


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
