"""Build the HIP model for an oracle ``TrunkCfg`` (the mirror image of tests/golden/make_golden.py's
``upstream_model``) and move case data to the GPU."""
from functools import partial

import numpy as np
import torch


def build_hip_model(cfg, share_embedding=True, norm_bias=False, learned_pos=()):
    from fourm.models import fm, fm_utils
    from fourm.models import encoder_embeddings as E
    from fourm.models import decoder_embeddings as Dm
    enc, dec, info = {}, {}, {}
    for m in cfg.mods:
        side = int(round(np.sqrt(m.n_pos))) if not m.is_seq else 0
        info[m.name] = {"id": m.id, "type": {"tok": "img", "patch": "img", "seq": "seq", "seq_emb": "seq_emb"}[m.kind], "max_tokens": m.n_pos}
        sincos = m.name not in learned_pos
        if m.in_enc:
            if m.kind == "tok":
                enc[m.name] = E.ImageTokenEncoderEmbedding(vocab_size=m.vocab, patch_size=m.patch, image_size=side * m.patch, sincos_pos_emb=sincos)
            elif m.kind == "patch":
                enc[m.name] = E.ImageEncoderEmbedding(num_channels=m.channels, patch_size=m.patch, image_size=side * m.patch)
            elif m.kind == "seq":
                enc[m.name] = E.SequenceEncoderEmbedding(vocab_size=m.vocab, max_length=m.n_pos, padding_idx=0)
            else:
                enc[m.name] = E.SequenceEmbEncoderEmbedding(max_length=m.n_pos, orig_emb_dim=m.orig_dim)
        if m.in_dec:
            if m.kind == "tok":
                dec[m.name] = Dm.ImageTokenDecoderEmbedding(vocab_size=m.vocab, patch_size=m.patch, image_size=side * m.patch,
                                                            sincos_pos_emb=sincos, share_embedding=share_embedding)
            else:
                dec[m.name] = Dm.SequenceDecoderEmbedding(vocab_size=m.vocab, max_length=m.n_pos, padding_idx=0, share_embedding=share_embedding)
    norm = partial(torch.nn.LayerNorm, eps=cfg.eps) if norm_bias else partial(fm_utils.LayerNorm, eps=cfg.eps, bias=False)
    return fm.FourM(encoder_embeddings=enc, decoder_embeddings=dec, modality_info=info, dim=cfg.dim, encoder_depth=cfg.enc_depth,
                    decoder_depth=cfg.dec_depth, num_heads=cfg.heads, mlp_ratio=cfg.mlp_ratio, qkv_bias=cfg.qkv_bias,
                    proj_bias=cfg.proj_bias, mlp_bias=cfg.mlp_bias, act_layer=torch.nn.SiLU if cfg.act == "silu" else torch.nn.GELU,
                    norm_layer=norm, gated_mlp=cfg.gated, qk_norm=cfg.qk_norm, decoder_causal_mask=cfg.causal,
                    decoder_sep_mask=cfg.sep, num_register_tokens=cfg.registers)


def to_device(mod_dict, device="cuda"):
    return {k: {a: b.to(device) for a, b in v.items()} for k, v in mod_dict.items()}


def tie(P, cfg, share):
    for m in cfg.mods:
        if m.in_enc and m.in_dec:
            P[f"decoder_embeddings.{m.name}.mod_emb"] = P[f"encoder_embeddings.{m.name}.mod_emb"]
        if m.in_dec and share:
            P[f"decoder_embeddings.{m.name}.to_logits.weight"] = P[f"decoder_embeddings.{m.name}.token_emb.weight"]
    return P
