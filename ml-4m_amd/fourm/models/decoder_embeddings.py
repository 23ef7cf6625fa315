"""Decoder-side modality embedders and output heads (parameter owners).

Names and signatures follow upstream ``fourm/models/decoder_embeddings.py`` (classes at :24, :156).
``to_logits.weight`` is tied to ``token_emb.weight`` when ``share_embedding`` (the training default).
The head GEMMs and the cross-entropy run grouped over all modalities in ``fourm.hip.engine``.
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from .encoder_embeddings import KIND_SEQ, KIND_TOK, _Embedder, _SeqPos
from .fm_utils import build_2d_sincos_posemb, pair


class _Head:
    def _make_head(self, padding_idx=None):
        self.token_emb = nn.Embedding(self.vocab_size, self.dim_tokens, padding_idx=padding_idx)
        self.to_logits = nn.Linear(self.dim_tokens, self.vocab_size, bias=False)
        if self.share_embedding:
            self.to_logits.weight = self.token_emb.weight

    def forward_embed(self, d):
        """Upstream contract: adds ``x``, ``emb`` and ``ids`` (= the token tensor) to ``d``.  Inference only."""
        from fourm.hip import functional
        d["x"], d["emb"] = functional.embed_modality(self, d, is_dec=True)
        d["ids"] = d["tensor"]
        return d

    def forward_logits(self, x: torch.Tensor) -> torch.Tensor:
        """Decoder states (..., D) -> logits (..., vocab); bf16 GEMM, returned in x's dtype."""
        from fourm.hip import functional
        return functional.linear(x, self.to_logits.weight, None)


class SequenceDecoderEmbedding(_Head, _SeqPos):
    """[upstream :24-152]"""
    kind = KIND_SEQ

    def __init__(self, vocab_size: int, max_length: int, dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True,
                 max_sincos_pos_emb: int = 512, padding_idx: int = 0, share_embedding: bool = True, **kwargs):
        super().__init__()
        self.vocab_size, self.max_length, self.dim_tokens = vocab_size, max_length, dim_tokens
        self.sincos_pos_emb, self.padding_idx, self.max_sincos_pos_emb = sincos_pos_emb, padding_idx, max_sincos_pos_emb
        self.share_embedding = share_embedding
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768, init_std=0.02):
        self.dim_tokens = dim_tokens
        self._seq_pos(init_std)
        self._make_head(self.padding_idx)


class ImageTokenDecoderEmbedding(_Head, _Embedder):
    """[upstream :156-268]"""
    kind = KIND_TOK

    def __init__(self, vocab_size: int, patch_size: Union[int, Tuple[int, int]] = 16, dim_tokens: Optional[int] = None,
                 sincos_pos_emb: bool = True, image_size: Union[int, Tuple[int]] = 224, share_embedding: bool = True, **kwargs):
        super().__init__()
        self.vocab_size, self.patch_size, self.dim_tokens = vocab_size, pair(patch_size), dim_tokens
        self.sincos_pos_emb, self.image_size, self.share_embedding = sincos_pos_emb, pair(image_size), share_embedding
        self.num_patches = (self.image_size[0] // self.patch_size[0]) * (self.image_size[1] // self.patch_size[1])
        if dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768, init_std=0.02):
        self.dim_tokens = dim_tokens
        h, w = self.image_size[0] // self.patch_size[0], self.image_size[1] // self.patch_size[1]
        self._make_pos(build_2d_sincos_posemb(h=h, w=w, embed_dim=dim_tokens) if self.sincos_pos_emb else None, h * w, init_std)
        self._make_head()


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
