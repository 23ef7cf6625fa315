"""The trunk sub-API generation / fine-tuning code calls directly (upstream generate.py:413-457,630-644,891-895):
``forward_encoder``, ``forward_decoder`` (no mask, dense causal mask), ``forward_logits``, stand-alone ``Block`` /
``DecoderBlock`` calls and ``VQ.tokens_to_embedding``, each against the CPU oracle run with bf16 rounding at the
autocast points.  Tolerance: relative Frobenius error, stated at each assert; measured values go to parity.jsonl."""
import random

import pytest
import torch

from oracle import fourm_oracle as O
from tests.golden.cases import build_case
from tests.parity_log import record
from tests.util_model import build_hip_model, tie

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def setup(name):
    case = build_case(name)
    model = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(case["sd"], strict=True)
    P = tie({k: v.clone() for k, v in case["sd"].items()}, case["cfg"], case["share_embedding"])
    return case, model.cuda().eval(), P


def rand_inputs(cfg, B, N, M, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, N, cfg.dim, generator=g)
    y = torch.randn(B, M, cfg.dim, generator=g)
    ctx = torch.randn(B, N, cfg.dim, generator=g)
    enc_mask = torch.zeros(B, 1, N, dtype=torch.bool)
    for b in range(B):
        enc_mask[b, 0, N - 1 - 2 * b:] = True            # a few padded keys per sample
    return x, y, ctx, enc_mask


@pytest.mark.parametrize("name,N,M", [("micro_swiglu", 20, 18), ("micro_gelu", 24, 12), ("ti_mod7", 128, 96)])
def test_forward_encoder_decoder(name, N, M):
    case, model, P = setup(name)
    cfg = case["cfg"]
    B = 2
    x, y, ctx, enc_mask = rand_inputs(cfg, B, N, M)
    num = O._Num(True)
    with torch.no_grad():
        want_e = O.encoder_forward(P, cfg, x, enc_mask, num)
        got_e = model.forward_encoder(x.cuda(), enc_mask.cuda())
        assert tuple(got_e.shape) == (B, N, cfg.dim) and got_e.dtype == torch.float32
        e_enc = rel(got_e, want_e)
        # decoder, no self-attention mask (generate.py:642) ...
        want_d = O.decoder_forward(P, cfg, y, ctx, enc_mask, None, num)
        got_d = model.forward_decoder(y.cuda(), ctx.cuda(), enc_mask.cuda(), None)
        e_dec = rel(got_d, want_d)
        # ... and an explicit causal (B, M, M) mask (generate.py:891-895)
        causal = torch.triu(torch.ones(M, M, dtype=torch.bool), 1)[None].expand(B, M, M)
        want_c = O.decoder_forward(P, cfg, y, ctx, enc_mask, causal, num)
        got_c = model.forward_decoder(y.cuda(), ctx.cuda(), enc_mask.cuda(), causal.cuda())
        e_cau = rel(got_c, want_c)
        assert rel(want_c, want_d) > 1e-2                                 # the mask matters in this test
    record("subapi.forward_encoder_decoder", case=name, enc=e_enc, dec_nomask=e_dec, dec_causal=e_cau)
    # bf16 pipeline vs bf16-emulating oracle over the full depth of the stack: 1e-2 relative Frobenius
    assert e_enc < 1e-2 and e_dec < 1e-2 and e_cau < 1e-2, (e_enc, e_dec, e_cau)


@pytest.mark.parametrize("name", ["micro_swiglu", "micro_gelu", "micro_qknorm"])
def test_standalone_blocks(name):
    """``blk(x, mask)`` / ``blk(x, context, sa_mask, xa_mask)`` the way upstream modules are called (fm_utils.py:331-366)."""
    case, model, P = setup(name)
    cfg = case["cfg"]
    B, N, M = 2, 20, 12
    x, y, ctx, enc_mask = rand_inputs(cfg, B, N, M, seed=1)
    num = O._Num(True)
    with torch.no_grad():
        pre = "encoder.1"
        want = x + O.self_attention(P, pre + ".attn", O._ln(P, pre + ".norm1", x, cfg, num), enc_mask, cfg, num)
        want = want + O.mlp(P, pre + ".mlp", O._ln(P, pre + ".norm2", want, cfg, num), cfg, num)
        got = model.encoder[1](x.cuda(), enc_mask.cuda())
        e1 = rel(got, want)
        pre = "decoder.0"
        causal = torch.triu(torch.ones(M, M, dtype=torch.bool), 1)[None].expand(B, M, M)
        w = y + O.self_attention(P, pre + ".self_attn", O._ln(P, pre + ".norm1", y, cfg, num), causal, cfg, num)
        w = w + O.cross_attention(P, pre + ".cross_attn", O._ln(P, pre + ".query_norm", w, cfg, num),
                                  O._ln(P, pre + ".context_norm", ctx, cfg, num), enc_mask, cfg, num)
        w = w + O.mlp(P, pre + ".mlp", O._ln(P, pre + ".norm2", w, cfg, num), cfg, num)
        got = model.decoder[0](y.cuda(), ctx.cuda(), sa_mask=causal.cuda(), xa_mask=enc_mask.cuda())
        e2 = rel(got, w)
    record("subapi.standalone_blocks", case=name, block=e1, decoder_block=e2)
    assert e1 < 6e-3 and e2 < 6e-3, (e1, e2)                               # one block: bf16 GEMM rounding only


@pytest.mark.parametrize("name", ["micro_swiglu", "ti_mod7"])
def test_forward_logits(name):
    """FourM.forward_logits: per-modality rows (y[mod_mask == id]) and return_all_logits   [fm.py:521-545]."""
    case, model, P = setup(name)
    cfg = case["cfg"]
    B, M = 2, 24
    g = torch.Generator().manual_seed(3)
    y = torch.randn(B, M, cfg.dim, generator=g)
    dec = [m for m in cfg.mods if m.in_dec]
    ids = torch.tensor([m.id for m in dec], dtype=torch.int16)
    mod_mask = ids[torch.randint(0, len(dec), (B, M), generator=g)]
    mod_mask[0, :3] = -1                                                  # padding rows belong to nobody
    dmd = {m.name: {} for m in dec}
    num = O._Num(True)
    with torch.no_grad():
        got = model.forward_logits(y.cuda(), dmd, mod_mask.cuda())
        allg = model.forward_logits(y.cuda(), dmd, mod_mask.cuda(), return_all_logits=True)
    worst = 0.0
    for m in dec:
        W = P[f"decoder_embeddings.{m.name}.to_logits.weight"]
        sel = mod_mask == m.id
        want = num.linear(y[sel], W, None)
        assert tuple(got[m.name].shape) == tuple(want.shape), m.name
        if want.numel():
            worst = max(worst, rel(got[m.name], want))
        want_all = num.linear(y, W, None)
        assert tuple(allg[m.name].shape) == (B, M, m.vocab)
        worst = max(worst, rel(allg[m.name], want_all))
    record("subapi.forward_logits", case=name, worst=worst)
    assert worst < 4e-3, worst                                             # one bf16 GEMM, bf16 output


def test_vq_tokens_to_embedding():
    """VQ.tokens_to_embedding = codebook rows, (B, h, w) -> (B, latent, h, w)   [vqvae.py:320-331]: exact."""
    from oracle import vq_oracle as V
    from tests.golden.cases import VQ_CASES
    from tests.test_vq import build
    c = VQ_CASES["vq_small"]
    cfg = V.vq_cfg(c["enc_type"], image=c["image"], patch=c["patch"], codebook=c["codebook"], post_mlp=c["post_mlp"])
    sd = V.seeded_vq_state_dict(cfg, seed=c["seed"])
    model = build(c, cfg)
    model.load_state_dict(sd)
    model = model.cuda().eval()
    tok = torch.randint(0, cfg.codebook, (3, cfg.grid, cfg.grid), generator=torch.Generator().manual_seed(0))
    emb = model.tokens_to_embedding(tok.cuda())
    want = sd["quantize._codebook.embed"][tok].permute(0, 3, 1, 2)
    assert tuple(emb.shape) == (3, cfg.latent, cfg.grid, cfg.grid)
    assert torch.equal(emb.cpu(), want)
