"""Modality registry: what "mod7" / "mod21" mean to the model (vocabularies, kinds, lengths, ids).

Same keys and field names as upstream ``fourm/data/modality_info.py:32-383`` for everything the model
side reads (``type``, ``vocab_size``, ``max_tokens``, ``id``, ``input_size``, ``patch_size``,
``encoder_embedding`` / ``decoder_embedding`` constructors).  The CPU data-transform half of the
upstream file (MODALITY_TRANSFORMS) belongs to the data loader and is out of scope here.
"""
from functools import partial

from fourm.models.decoder_embeddings import ImageTokenDecoderEmbedding, SequenceDecoderEmbedding
from fourm.models.encoder_embeddings import (ImageEncoderEmbedding, ImageTokenEncoderEmbedding,
                                             SequenceEmbEncoderEmbedding, SequenceEncoderEmbedding)
from fourm.utils.misc import generate_uint15_hash


def _grid_tokens(name, vocab, patch=16, res=224, **extra):
    """Pre-tokenized image-like modality at a fixed resolution."""
    info = dict(input_size=res, patch_size=patch, vocab_size=vocab,
                encoder_embedding=partial(ImageTokenEncoderEmbedding, vocab_size=vocab),
                decoder_embedding=partial(ImageTokenDecoderEmbedding, vocab_size=vocab),
                min_tokens=0, max_tokens=None, type="img", id=generate_uint15_hash(name), pretokenized=True)
    info.update(extra)
    return info


def _global_tokens(name, vocab):
    """16 global feature tokens with learned positions (patch 56 over a 224 'image')."""
    return dict(vocab_size=vocab, patch_size=56,
                encoder_embedding=partial(ImageTokenEncoderEmbedding, vocab_size=vocab, sincos_pos_emb=False),
                decoder_embedding=partial(ImageTokenDecoderEmbedding, vocab_size=vocab, sincos_pos_emb=False),
                min_tokens=0, max_tokens=16, type="img", id=generate_uint15_hash(name), pretokenized=True)


def _pixels(name, res):
    return dict(input_size=res, patch_size=16, encoder_embedding=partial(ImageEncoderEmbedding, num_channels=3),
                decoder_embedding=None, min_tokens=0, max_tokens=None, type="img", num_channels=3,
                id=generate_uint15_hash(name), path="rgb")


def _sequence(name, max_length, max_tokens=None, vocab=30_000, **extra):
    info = dict(vocab_size=vocab,
                encoder_embedding=partial(SequenceEncoderEmbedding, vocab_size=vocab, max_length=max_length, padding_idx=0),
                decoder_embedding=partial(SequenceDecoderEmbedding, vocab_size=vocab, max_length=max_length, padding_idx=0),
                min_tokens=0, max_tokens=max_tokens if max_tokens is not None else max_length, type="seq",
                id=generate_uint15_hash(name))
    info.update(extra)
    return info


def _raw(name, kind, **fields):
    """Un-tokenized source modality (only used when training tokenizers): no embeddings."""
    return dict(type=kind, id=generate_uint15_hash(name), **fields)


MODALITY_INFO = {
    # ---- 4M-7 -------------------------------------------------------------------------------
    "rgb@224": _pixels("rgb@224", 224),
    "rgb": _raw("rgb", "img", num_channels=3, path="rgb"),
    "caption": _sequence("caption", 256),
    "det": _sequence("det", 256),
    "tok_rgb@224": _grid_tokens("tok_rgb@224", 16384),
    "tok_depth@224": _grid_tokens("tok_depth@224", 8192),
    "depth": _raw("depth", "img", num_channels=1),
    "tok_normal@224": _grid_tokens("tok_normal@224", 8192),
    "normal": _raw("normal", "img", num_channels=3),
    "tok_semseg@224": _grid_tokens("tok_semseg@224", 4096),
    "semseg_coco": _raw("semseg_coco", "img", num_channels=64, num_labels=134),
    "tok_clip@224": _grid_tokens("tok_clip@224", 8192),
    "CLIP-B16": _raw("CLIP-B16", "feature_map", num_channels=512),
    # ---- 4M-21 ------------------------------------------------------------------------------
    "t5_caption": dict(encoder_embedding=partial(SequenceEmbEncoderEmbedding, max_length=77, padding_idx=0),
                       decoder_embedding=None, min_tokens=0, max_tokens=77, type="seq_emb",
                       id=generate_uint15_hash("t5_caption")),
    "metadata": _sequence("metadata", 40, shared_vocab=["caption"], path="metadata"),
    "human_poses": _sequence("human_poses", 263, max_tokens=275, num_channels=207, shared_vocab=["caption"]),
    "color_palette": _sequence("color_palette", 23, shared_vocab=["caption"], path="color_palette"),
    "sam_mask": dict(encoder_embedding=None, decoder_embedding=None, min_tokens=0, max_tokens=64, type="img",
                     num_channels=1, id=generate_uint15_hash("sam_mask")),
    "sam_instance": _sequence("sam_instance", 290, shared_vocab=["caption"], pretokenized=True),
    "tok_canny_edge@224": _grid_tokens("tok_canny_edge@224", 8192),
    "canny_edge": _raw("canny_edge", "img", num_channels=1),
    "tok_sam_edge@224": _grid_tokens("tok_sam_edge@224", 8192),
    "tok_dinov2@224": _grid_tokens("tok_dinov2@224", 8192, patch=14),
    "DINOv2-B14": _raw("DINOv2-B14", "feature_map", num_channels=768),
    "tok_imagebind@224": _grid_tokens("tok_imagebind@224", 8192, patch=14),
    "ImageBind-H14": _raw("ImageBind-H14", "feature_map", num_channels=1280),
    "tok_dinov2_global": _global_tokens("tok_dinov2_global", 8192),
    "DINOv2-B14-global": _raw("DINOv2-B14-global", "feature_map", num_channels=768),
    "tok_imagebind_global": _global_tokens("tok_imagebind_global", 8192),
    "ImageBind-H14-global": _raw("ImageBind-H14-global", "feature_map", num_channels=1280),
    # ---- 224 -> 448 super-resolution ----------------------------------------------------------
    "rgb@448": _pixels("rgb@448", 448),
    "tok_rgb@448": _grid_tokens("tok_rgb@448", 16384, res=448),
    "tok_depth@448": _grid_tokens("tok_depth@448", 8192, res=448),
    "tok_normal@448": _grid_tokens("tok_normal@448", 8192, res=448),
    "tok_semseg@448": _grid_tokens("tok_semseg@448", 4096, res=448),
    "tok_clip@448": _grid_tokens("tok_clip@448", 8192, res=448),
}


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
