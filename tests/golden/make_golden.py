#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the UNMODIFIED upstream model.

Runs only where /root/reference exists (the build container).  For every case it
  1. builds the upstream ``fourm.models.fm.FourM`` with upstream embedding modules,
  2. loads the deterministic weights of ``oracle.fourm_oracle.seeded_state_dict`` (strict: proves the
     state_dict key/shape layout the HIP package must reproduce),
  3. runs forward (+ backward) on the deterministic batch of ``synthetic_mod_dict``,
  4. checks the oracle restatement against it (fp32: bit-exact integers, ~1e-5 floats),
  5. writes ``<case>.npz`` with the upstream outputs.  Weights/inputs are *not* stored: they are
     regenerated from the same seeded generators on the test side; checksums guard against RNG drift.

    python tests/golden/make_golden.py            # (re)write fixtures
    python tests/golden/make_golden.py --check    # compare upstream vs oracle only
"""
import argparse
import os
import random
import sys
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_stubs  # noqa: E402

ref_stubs.install()

from fourm.models import fm as ref_fm  # noqa: E402
from fourm.models import fm_utils as ref_utils  # noqa: E402
from fourm.models import encoder_embeddings as ref_enc  # noqa: E402
from fourm.models import decoder_embeddings as ref_dec  # noqa: E402

from oracle import fourm_oracle as O  # noqa: E402
from tests.golden.cases import CASES, build_case  # noqa: E402


def upstream_model(cfg: O.TrunkCfg, share_embedding: bool, norm_bias: bool, learned_pos):
    enc, dec, info = {}, {}, {}
    for m in cfg.mods:
        side = int(round(np.sqrt(m.n_pos))) if not m.is_seq else 0
        info[m.name] = {"id": m.id, "type": {"tok": "img", "patch": "img", "seq": "seq", "seq_emb": "seq_emb"}[m.kind]}
        sincos = m.name not in learned_pos
        if m.in_enc:
            if m.kind == "tok":
                enc[m.name] = ref_enc.ImageTokenEncoderEmbedding(vocab_size=m.vocab, patch_size=m.patch,
                                                                 image_size=side * m.patch, sincos_pos_emb=sincos)
            elif m.kind == "patch":
                enc[m.name] = ref_enc.ImageEncoderEmbedding(num_channels=m.channels, patch_size=m.patch,
                                                            image_size=side * m.patch)
            elif m.kind == "seq":
                enc[m.name] = ref_enc.SequenceEncoderEmbedding(vocab_size=m.vocab, max_length=m.n_pos, padding_idx=0)
            else:
                enc[m.name] = ref_enc.SequenceEmbEncoderEmbedding(max_length=m.n_pos, orig_emb_dim=m.orig_dim)
        if m.in_dec:
            if m.kind == "tok":
                dec[m.name] = ref_dec.ImageTokenDecoderEmbedding(vocab_size=m.vocab, patch_size=m.patch,
                                                                 image_size=side * m.patch, sincos_pos_emb=sincos,
                                                                 share_embedding=share_embedding)
            else:
                dec[m.name] = ref_dec.SequenceDecoderEmbedding(vocab_size=m.vocab, max_length=m.n_pos, padding_idx=0,
                                                               share_embedding=share_embedding)
    if norm_bias:
        norm = partial(torch.nn.LayerNorm, eps=cfg.eps)
    else:
        norm = partial(ref_utils.LayerNorm, eps=cfg.eps, bias=False)
    model = ref_fm.FourM(
        encoder_embeddings=enc, decoder_embeddings=dec, modality_info=info, dim=cfg.dim,
        encoder_depth=cfg.enc_depth, decoder_depth=cfg.dec_depth, num_heads=cfg.heads, mlp_ratio=cfg.mlp_ratio,
        qkv_bias=cfg.qkv_bias, proj_bias=cfg.proj_bias, mlp_bias=cfg.mlp_bias,
        act_layer=torch.nn.SiLU if cfg.act == "silu" else torch.nn.GELU, norm_layer=norm, gated_mlp=cfg.gated,
        qk_norm=cfg.qk_norm, decoder_causal_mask=cfg.causal, decoder_sep_mask=cfg.sep,
        num_register_tokens=cfg.registers)
    return model


def clone_mod_dict(md):
    return {k: {a: b.clone() for a, b in v.items()} for k, v in md.items()}


def dec_order_for_seed(names, seed):
    """The order upstream's ``random.sample(mod_dict.items(), n)`` (fm.py:306) produces."""
    random.seed(seed)
    return random.sample(list(names), len(names))


def run_case(name: str, check_only: bool):
    case = build_case(name)
    cfg, sd, md = case["cfg"], case["sd"], case["mod_dict"]
    N, M, loss_type, seed = case["N"], case["M"], case["loss_type"], case["order_seed"]
    model = upstream_model(cfg, case["share_embedding"], case["norm_bias"], case["learned_pos"])
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    ref_keys = list(model.state_dict().keys())
    assert sorted(ref_keys) == sorted(sd.keys()), "state_dict layout mismatch"
    model.train()
    big = case.get("big", False)
    if big:       # batch 256: recompute each block in the backward instead of keeping ~70 GB of activations (identical arithmetic)
        from torch.utils.checkpoint import checkpoint
        for blk in list(model.encoder) + list(model.decoder):
            blk.forward = (lambda f: (lambda *a, **k: checkpoint(f, *a, use_reentrant=False, **k)))(blk.forward)

    dec_names = [n for n in md if n in model.decoder_embeddings]
    order = dec_order_for_seed(dec_names, seed)

    # --- upstream: selection outputs
    d1 = clone_mod_dict(md)
    enc_md = {m: model.encoder_embeddings[m](d) for m, d in d1.items() if m in model.encoder_embeddings}
    e_tok, e_emb, e_mask, e_mod = model.forward_mask_encoder(enc_md, N)
    dec_md = {m: model.decoder_embeddings[m].forward_embed(d) for m, d in d1.items() if m in model.decoder_embeddings}
    random.seed(seed)
    d_tok, d_emb, d_mask, d_tgt, d_attn, d_mod = model.forward_mask_decoder(dec_md, M)

    # --- upstream: loss + grads, logits
    random.seed(seed)
    model.zero_grad()
    loss, mod_loss = model(clone_mod_dict(md), N, M, loss_type=loss_type)
    loss.sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    random.seed(seed)
    with torch.no_grad():
        logits = model(clone_mod_dict(md), N, M, return_logits=True)

    if big:
        _write(name, model, sd, md, order, ref_keys, cfg, e_tok, e_emb, e_mask, e_mod, d_tok, d_emb, d_mask, d_mod, d_tgt, d_attn, loss, mod_loss, logits, grads)
        return
    # --- oracle on the same weights / inputs
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    # tied tensors must stay tied for gradient accumulation
    for m in cfg.mods:
        if m.in_enc and m.in_dec:
            P[f"decoder_embeddings.{m.name}.mod_emb"] = P[f"encoder_embeddings.{m.name}.mod_emb"]
        if m.in_dec and case["share_embedding"]:
            P[f"decoder_embeddings.{m.name}.to_logits.weight"] = P[f"decoder_embeddings.{m.name}.token_emb.weight"]
    taps = {}
    o_loss, o_mod = O.fourm_forward(P, cfg, clone_mod_dict(md), N, M, order, loss_type=loss_type, taps=taps)
    o_loss.sum().backward()
    with torch.no_grad():
        o_logits = O.fourm_forward(P, cfg, clone_mod_dict(md), N, M, order, return_logits=True)

    def same_int(a, b, what):
        assert torch.equal(a.long(), b.long()), f"{name}: {what} differs"

    def close(a, b, what, tol=2e-5):
        err = (a - b).abs().max().item() / max(1e-6, b.abs().max().item())
        assert err < tol, f"{name}: {what} rel-max err {err:.3e}"
        return err

    same_int(taps["enc_mask"], e_mask, "encoder mask"); same_int(taps["enc_mod_mask"], e_mod, "encoder mod_mask")
    same_int(taps["dec_mask"], d_mask, "decoder mask"); same_int(taps["dec_mod_mask"], d_mod, "decoder mod_mask")
    same_int(taps["dec_target_ids"], d_tgt, "target ids"); same_int(taps["dec_attn_mask"], d_attn, "decoder attn mask")
    # gathers are exact; the dense projections (pixels, T5 rows) only agree to fp32 summation order
    close(taps["enc_tokens"], e_tok.float().detach(), "encoder tokens", 1e-5)
    assert torch.equal(taps["enc_emb"], e_emb), "encoder emb"
    assert torch.equal(taps["dec_tokens"], d_tok), "decoder tokens"
    assert torch.equal(taps["dec_emb"], d_emb), "decoder emb"
    errs = {"loss": close(o_loss.reshape(-1), loss.reshape(-1).detach(), "loss")}
    for k in mod_loss:
        close(o_mod[k].reshape(-1), mod_loss[k].reshape(-1).detach(), f"mod loss {k}")
    for k in logits:
        errs[f"logits/{k}"] = close(o_logits[k], logits[k], f"logits {k}", 1e-4)
    gmax = 0.0
    for k, g in grads.items():
        og = P[k].grad
        assert og is not None, f"oracle has no grad for {k}"
        gmax = max(gmax, close(og, g, f"grad {k}", 2e-4))
    errs["grad_max"] = gmax
    print(f"[{name}] upstream==oracle  loss={loss.sum().item():.6f}  order={order}  "
          f"max rel err: loss {errs['loss']:.1e} grads {gmax:.1e}")
    if check_only:
        return
    _write(name, model, sd, md, order, ref_keys, cfg, e_tok, e_emb, e_mask, e_mod, d_tok, d_emb, d_mask, d_mod, d_tgt, d_attn, loss, mod_loss, logits, grads)


def _write(name, model, sd, md, order, ref_keys, cfg, e_tok, e_emb, e_mask, e_mod, d_tok, d_emb, d_mask, d_mod, d_tgt, d_attn, loss, mod_loss, logits, grads):
    out = {
        "meta/order": np.array(order), "meta/keys": np.array(ref_keys),
        "meta/shapes": np.array([",".join(map(str, model.state_dict()[k].shape)) for k in ref_keys]),
        "meta/param_keys": np.array([k for k, _ in model.named_parameters()]),
        "meta/buffer_keys": np.array([k for k, _ in model.named_buffers()]),
        "meta/weight_checksum": np.array(sum(float(v.double().abs().sum()) for v in sd.values())),
        "meta/input_checksum": np.array(sum(float(t.double().abs().sum()) for d in md.values() for t in d.values())),
        "enc/mask": e_mask.numpy(), "enc/mod_mask": e_mod.numpy(),
        "dec/mask": d_mask.numpy(), "dec/mod_mask": d_mod.numpy(), "dec/target_ids": d_tgt.numpy(),
        "dec/attn_mask": np.packbits(d_attn.numpy(), axis=-1),
        "loss": loss.detach().reshape(-1).numpy(),
    }
    small = cfg.dim <= 128
    if small:
        out.update({"enc/tokens": e_tok.detach().float().numpy(), "enc/emb": e_emb.detach().numpy(),
                    "dec/tokens": d_tok.detach().numpy(), "dec/emb": d_emb.detach().numpy()})
    else:
        out.update({"enc/tokens_rowsum": e_tok.detach().float().sum(-1).numpy(), "enc/emb_rowsum": e_emb.detach().sum(-1).numpy(),
                    "dec/tokens_rowsum": d_tok.detach().sum(-1).numpy(), "dec/emb_rowsum": d_emb.detach().sum(-1).numpy()})
    for k, v in mod_loss.items():
        out[f"mod_loss/{k}"] = v.detach().reshape(-1).numpy()
    for k, v in logits.items():
        out[f"logits_fro/{k}"] = np.array(float(v.double().norm()))
        out[f"logits_head/{k}"] = v[:, :8, :16].numpy() if not small else v.numpy().astype(np.float32)
    for k, g in grads.items():
        out[f"grad_l2/{k}"] = np.array(float(g.double().norm()))
        flat = g.reshape(-1)
        out[f"grad_head/{k}"] = flat[:16].numpy()
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"    wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    ap.add_argument("cases", nargs="*", default=list(CASES))
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    for c in a.cases:
        run_case(c, a.check)
