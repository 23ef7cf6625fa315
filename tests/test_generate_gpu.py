"""Generation step: the sampling kernels bit-exact against the numpy oracle, and one MaskGIT step of
``GenerationSampler`` (upstream generate.py:628-661) against the CPU oracle pipeline."""
import numpy as np
import pytest
import torch

from oracle import fourm_oracle as O
from oracle import sample_oracle as S
from tests.golden.cases import build_case
from tests.parity_log import record
from tests.util_model import build_hip_model, tie

pytestmark = pytest.mark.gpu


def sampler(model=None):
    from fourm.models.generate import GenerationSampler
    return GenerationSampler(model)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("V", [64, 1000, 16384, 30000])
def test_sample_tokens_bit_exact(V, dtype):
    """Same logits + same uniforms -> the same token ids as oracle/sample_oracle.py, bit for bit, for every filter combination."""
    g = torch.Generator().manual_seed(V)
    R = 48
    logits = (torch.randn(R, V, generator=g) * 2.5).to(dtype)
    logits[3] = logits[3, 0]                                   # a constant row: everything ties
    logits[5, 17] = 40.0                                       # a row dominated by one entry
    u = torch.rand(R, generator=g)
    u[0], u[1] = 0.0, 0.99999994                               # the ends of the CDF
    smp = sampler()
    lf = logits.float().numpy()
    bad = []
    for temperature, top_k, top_p in [(1.0, 0, 0.0), (0.7, 50, 0.0), (1.0, 0, 0.9), (0.5, 100, 0.8), (1.3, 0.01, 0.95), (0.0, 0, 0.0), (1.0, 1, 0.0)]:
        ids, probs = smp.sample_tokens(logits.cuda(), temperature, top_k, top_p, uniforms=u.cuda())
        k = smp._top_k_int(top_k, V)
        want_ids, want_p = S.sample_tokens(lf, temperature, k, top_p, u.numpy())
        if not np.array_equal(ids.cpu().numpy(), want_ids):
            bad.append(("ids", temperature, top_k, top_p, np.nonzero(ids.cpu().numpy() != want_ids)[0][:5].tolist()))
        if not np.array_equal(probs.cpu().numpy(), want_p):
            bad.append(("probs", temperature, top_k, top_p, float(np.abs(probs.cpu().numpy() - want_p).max())))
    assert not bad, bad
    # statistics: frequencies follow softmax(logits / T) restricted to the nucleus
    row = logits[7:8].float()
    rep = row.repeat(8192, 1).cuda()
    ids, _ = smp.sample_tokens(rep, 1.0, 20, 0.0, generator=torch.Generator(device="cuda").manual_seed(1))
    keep = S.survivors(row[0].numpy(), 20, 0.0)
    p = torch.softmax(row[0].masked_fill(~torch.from_numpy(keep), float("-inf")), -1)
    freq = torch.bincount(ids.cpu(), minlength=V).float() / 8192
    assert float((freq - p).abs().max()) < 0.03 and float(freq[~torch.from_numpy(keep)].sum()) == 0.0


def gen_mod_dict(cfg, B, target, seed=0, cond_tokens=24):
    """Conditioning: ``cond_tokens`` visible inputs spread over the other modalities; target modality fully masked, to be decoded."""
    md = O.synthetic_mod_dict(cfg, B, cond_tokens, 0, seed=seed, no_target=tuple(m.name for m in cfg.mods))
    for name, d in md.items():
        d["target_mask"][:] = True
    t = md[target]
    t["input_mask"][:] = True
    t["target_mask"][:] = False
    return md


@pytest.mark.parametrize("case_name,target", [("micro_swiglu", "tok_a@32"), ("ti_mod7", "tok_depth@224")])
def test_maskgit_step(case_name, target):
    case = build_case(case_name)
    cfg = case["cfg"]
    model = build_hip_model(cfg, case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(case["sd"], strict=True)
    model = model.cuda().eval()
    P = tie({k: v.clone() for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    B = 3
    md = gen_mod_dict(cfg, B, target)
    dev_md = {k: {a: b.cuda() for a, b in v.items()} for k, v in md.items()}
    smp = sampler(model)
    logits, mod_pos = smp.forward_enc_dec_maskgit_batched(dev_md, target)
    spec = cfg.mod(target)
    Npos = spec.n_pos
    assert tuple(logits.shape) == (B, Npos, spec.vocab) and torch.equal(mod_pos.cpu(), torch.arange(Npos, dtype=torch.int32)[None].expand(B, -1))
    # ---- oracle: the same forward from the restated pieces (bf16 rounding at the autocast points) ----
    num = O._Num(True)
    with torch.no_grad():
        n_enc = max(int(sum((~md[m.name]["input_mask"].reshape(B, -1)[b]).sum() for m in cfg.mods if m.in_enc)) for b in range(B))
        enc = O.select_encoder(P, cfg, md, n_enc, num)
        x = O.encoder_forward(P, cfg, enc["tokens"] + enc["emb"], enc["mask"], num)
        ctx = num.linear(x, P["decoder_proj_context.weight"], P["decoder_proj_context.bias"]) + enc["emb"]
        _, e, _ = O.embed_decoder_modality(P, spec, md[target])
        y0 = P["mask_token"].expand(B, Npos, -1) + e.float()
        y = O.decoder_forward(P, cfg, y0, ctx, enc["mask"], None, num)
        want = num.linear(y, P[f"decoder_embeddings.{target}.to_logits.weight"], None)
    err = float((logits.float().cpu() - want).norm() / want.norm())
    record("generate.maskgit_logits", case=case_name, rel=err)
    assert err < 1.2e-2, err                                     # two bf16 pipelines (see test_model_gpu.LOGIT_BOUNDS)
    # ---- sampling + commit: bit-exact given the kernel's own logits and the same uniforms ----
    u = torch.rand(B * Npos, generator=torch.Generator().manual_seed(5))
    num_select = max(1, Npos // 4)
    lf = logits.float().cpu().numpy().reshape(B * Npos, -1).copy()
    before = {k: v.clone() for k, v in dev_md[target].items()}
    smp.maskgit_step_batched(dev_md, target, num_select, 0.8, 30, 0.9, uniforms=u.cuda())
    ids, probs = S.sample_tokens(lf, 0.8, 30, 0.9, u.numpy())
    st = smp.last_step
    assert np.array_equal(st["samples"].cpu().numpy().reshape(-1), ids)
    t = before["tensor"].cpu().numpy().reshape(B, -1).copy()
    im, tmk = before["input_mask"].cpu().numpy().reshape(B, -1).copy(), before["target_mask"].cpu().numpy().reshape(B, -1).copy()
    top = S.maskgit_commit(probs.reshape(B, Npos), ids.reshape(B, Npos), mod_pos.cpu().numpy(), num_select, t, im, tmk)
    assert np.array_equal(st["top_indices"].cpu().numpy(), top)
    assert np.array_equal(dev_md[target]["tensor"].cpu().numpy().reshape(B, -1), t)
    assert np.array_equal(dev_md[target]["input_mask"].cpu().numpy().reshape(B, -1), im)
    assert np.array_equal(dev_md[target]["target_mask"].cpu().numpy().reshape(B, -1), tmk)
    assert int(tmk.sum()) == B * num_select and int((~im).sum()) == B * num_select
    # ---- the whole schedule: every position decoded after the last step, ids inside the vocabulary ----
    smp.generate_maskgit(dev_md, target, num_steps=3, temperature=1.0, top_k=0, top_p=0.0, generator=torch.Generator(device="cuda").manual_seed(2))
    assert bool(dev_md[target]["target_mask"].all()) and not bool(dev_md[target]["input_mask"].any())
    tk = dev_md[target]["tensor"]
    assert int(tk.min()) >= 0 and int(tk.max()) < spec.vocab


def _oracle_logits(P, cfg, md, target, positions, B):
    """Oracle forward of one generation step decoding ``positions`` (same for every sample) of ``target``."""
    num = O._Num(True)
    spec = cfg.mod(target)
    with torch.no_grad():
        n_enc = max(int(sum((~md[m.name]["input_mask"].reshape(B, -1)[b]).sum() for m in cfg.mods if m.in_enc)) for b in range(B))
        enc = O.select_encoder(P, cfg, md, n_enc, num)
        x = O.encoder_forward(P, cfg, enc["tokens"] + enc["emb"], enc["mask"], num)
        ctx = num.linear(x, P["decoder_proj_context.weight"], P["decoder_proj_context.bias"]) + enc["emb"]
        _, e, _ = O.embed_decoder_modality(P, spec, md[target])
        e = e.float()[:, torch.as_tensor(positions, dtype=torch.long)]
        y = O.decoder_forward(P, cfg, P["mask_token"].expand(B, len(positions), -1) + e, ctx, enc["mask"], None, num)
        return num.linear(y, P[f"decoder_embeddings.{target}.to_logits.weight"], None)


def test_roar_and_guided_steps():
    """ROAR (random order) step and the classifier-free-guided MaskGIT / ROAR steps (generate.py:481-514, :665-703, :745-816): the
    decoded positions equal the restated argsort rule, the logits follow the oracle pipeline on those positions, guidance combines the
    two passes in fp32 exactly, sampling + commit are bit-exact given the kernel's logits and the same uniforms."""
    case_name, target = "ti_mod7", "tok_depth@224"
    case = build_case(case_name)
    cfg = case["cfg"]
    model = build_hip_model(cfg, case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(case["sd"], strict=True)
    model = model.cuda().eval()
    P = tie({k: v.clone() for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    B, spec = 2, cfg.mod(target)
    Npos = spec.n_pos
    md = gen_mod_dict(cfg, B, target, seed=3, cond_tokens=40)
    dev = lambda d: {k: {a: b.cuda() for a, b in v.items()} for k, v in d.items()}
    smp = sampler(model)
    # ---- ROAR ----
    dev_md = dev(md)
    noise = torch.rand(Npos, generator=torch.Generator().manual_seed(9))
    n_sel = 40
    want_pos = S.roar_positions(md[target]["target_mask"].reshape(B, -1).numpy(), noise.numpy(), n_sel)
    logits, mod_pos = smp.forward_enc_dec_roar_batched(dev_md, target, n_sel, order_noise=noise.cuda())
    assert np.array_equal(mod_pos.cpu().numpy(), want_pos) and tuple(logits.shape) == (B, n_sel, spec.vocab)
    assert np.array_equal(want_pos[0], want_pos[1])
    want = _oracle_logits(P, cfg, md, target, want_pos[0], B)
    err = float((logits.float().cpu() - want).norm() / want.norm())
    record("generate.roar_logits", case=case_name, rel=err)
    assert err < 1.2e-2, err
    u = torch.rand(B * n_sel, generator=torch.Generator().manual_seed(6))
    lf = logits.float().cpu().numpy().reshape(B * n_sel, -1).copy()
    smp.roar_step_batched(dev_md, target, n_sel, 1.0, 0, 0.95, uniforms=u.cuda(), order_noise=noise.cuda())
    ids, _ = S.sample_tokens(lf, 1.0, 0, 0.95, u.numpy())
    got_t = dev_md[target]["tensor"].cpu().numpy().reshape(B, -1)
    tmk, im = dev_md[target]["target_mask"].cpu().numpy().reshape(B, -1), dev_md[target]["input_mask"].cpu().numpy().reshape(B, -1)
    for b in range(B):
        assert np.array_equal(got_t[b, want_pos[b]], ids.reshape(B, n_sel)[b])
        chosen = np.zeros(Npos, dtype=bool); chosen[want_pos[b]] = True
        assert np.array_equal(tmk[b], chosen) and np.array_equal(~im[b], chosen)          # exactly the chosen positions were committed
    # a second step decodes n_sel OTHER positions
    smp.roar_step_batched(dev_md, target, n_sel, 1.0, 0, 0.95, generator=torch.Generator(device="cuda").manual_seed(1))
    assert int(dev_md[target]["target_mask"].sum()) == 2 * B * n_sel
    # ---- classifier-free guidance ----
    cond_mod = next(m.name for m in cfg.mods if m.in_enc and m.name != target and m.kind == "tok"
                    and bool((~md[m.name]["input_mask"]).any()))
    dev_md = dev(md)
    unc_md = smp.unconditional_dict(dev_md, [cond_mod])
    assert bool(unc_md[cond_mod]["input_mask"].all()) and not bool(unc_md[cond_mod]["target_mask"].any())       # empty_img_modality
    assert bool((~dev_md[cond_mod]["input_mask"]).any())                                                     # ... on a copy
    lc, _ = smp.forward_enc_dec_maskgit_batched(dev_md, target); lc = lc.float().cpu().numpy()
    lu, _ = smp.forward_enc_dec_maskgit_batched(unc_md, target); lu = lu.float().cpu().numpy()
    assert float(np.abs(lc - lu).max()) > 1e-3                                                                # the conditioning matters
    g, pos = smp._guided_logits(dev_md, target, [cond_mod], 2.5)
    want_g = S.cfg_logits(lc, lu, 2.5)
    assert np.array_equal(g.cpu().numpy(), want_g)
    u = torch.rand(B * Npos, generator=torch.Generator().manual_seed(8))
    k_sel = 30
    before = {k: v.clone() for k, v in dev_md[target].items()}
    smp.guided_maskgit_step_batched(dev_md, target, k_sel, 0.7, 50, 0.0, conditioning=[cond_mod], guidance_scale=2.5, uniforms=u.cuda())
    ids, probs = S.sample_tokens(want_g.reshape(B * Npos, -1), 0.7, 50, 0.0, u.numpy())
    t = before["tensor"].cpu().numpy().reshape(B, -1).copy()
    im, tmk = before["input_mask"].cpu().numpy().reshape(B, -1).copy(), before["target_mask"].cpu().numpy().reshape(B, -1).copy()
    S.maskgit_commit(probs.reshape(B, Npos), ids.reshape(B, Npos), pos.cpu().numpy(), k_sel, t, im, tmk)
    assert np.array_equal(dev_md[target]["tensor"].cpu().numpy().reshape(B, -1), t)
    assert np.array_equal(dev_md[target]["target_mask"].cpu().numpy().reshape(B, -1), tmk)
    # several weighted conditions: l_u + sum_i w_i (l_i - l_u) of three separate forwards, then the same sample + commit on every dict
    md_m = dev(md)
    cond2 = next(m.name for m in cfg.mods if m.in_enc and m.name not in (target, cond_mod) and m.kind == "tok" and bool((~md[m.name]["input_mask"]).any()))
    unc = smp.unconditional_dict(md_m, [cond_mod, cond2])
    c1, c2 = smp.unconditional_dict(md_m, [cond2]), smp.unconditional_dict(md_m, [cond_mod])
    l1, _ = smp.forward_enc_dec_maskgit_batched(c1, target); l1 = l1.float().cpu().numpy()
    l2, _ = smp.forward_enc_dec_maskgit_batched(c2, target); l2 = l2.float().cpu().numpy()
    l0, _ = smp.forward_enc_dec_maskgit_batched(unc, target); l0 = l0.float().cpu().numpy()
    gm, _ = smp._multi_guided_logits(unc, [c1, c2], [1.5, 0.5], target)
    want_m = l0 + (np.float32(1.5) * (l1 - l0) + np.float32(0.5) * (l2 - l0))
    assert np.allclose(gm.cpu().numpy(), want_m, rtol=0, atol=2e-5 * float(np.abs(want_m).max()))
    unc, conds = smp.multi_guided_maskgit_step_batched(unc, [c1, c2], [1.5, 0.5], target, k_sel, 1.0, 0, 0.9, seed=2)
    assert int(unc[target]["target_mask"].sum()) == B * k_sel
    for cd in conds:
        assert torch.equal(cd[target]["tensor"], unc[target]["tensor"]) and torch.equal(cd[target]["target_mask"], unc[target]["target_mask"])
    # guided ROAR: runs, commits exactly n_sel positions per sample
    smp.guided_roar_step_batched(dev_md, target, n_sel, 1.0, 0, 0.9, conditioning=[cond_mod], guidance_scale=1.5,
                                 generator=torch.Generator(device="cuda").manual_seed(4))
    assert int(dev_md[target]["target_mask"].sum()) == B * (k_sel + n_sel)


@pytest.mark.parametrize("case_name,target", [("micro_swiglu", None), ("ti_mod7", "caption")])
def test_autoregressive_kv_cache(case_name, target):
    """Autoregressive decoding of a sequence modality with the K/V cache (upstream generate.py:850-914 recomputes the whole prefix per
    token): the logits of every step follow the oracle's NON-cached recompute on the same prefix (full decoder under a causal mask),
    and the sampled ids are bit-exact given the kernel's logits and the same uniforms."""
    case = build_case(case_name)
    cfg = case["cfg"]
    if target is None:
        target = next(m.name for m in cfg.mods if m.kind == "seq" and m.in_dec)
    spec = cfg.mod(target)
    model = build_hip_model(cfg, case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(case["sd"], strict=True)
    model = model.cuda().eval()
    P = tie({k: v.clone() for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    B, n_prompt, n_gen = 3, 2, 9
    md = O.synthetic_mod_dict(cfg, B, 30, 0, seed=11, no_target=tuple(m.name for m in cfg.mods))
    for name, d in md.items():
        d["target_mask"][:] = True
    t = md[target]
    g = torch.Generator().manual_seed(12)
    t["tensor"] = torch.randint(5, spec.vocab, t["tensor"].shape, generator=g, dtype=t["tensor"].dtype)
    t["input_mask"][:] = True; t["input_mask"][:, :n_prompt] = False          # a short visible prompt ...
    t["target_mask"][:] = True; t["target_mask"][:, n_prompt:n_prompt + n_gen] = False      # ... then the positions to generate
    dev_md = {k: {a: b.cuda() for a, b in v.items()} for k, v in md.items()}
    smp = sampler(model)
    steps = n_gen
    u = torch.rand(steps, B, generator=torch.Generator().manual_seed(13))
    out = smp.autoregressive_generate(dev_md, target, temperature=0.9, top_k=40, top_p=0.0, use_eos=False, uniforms=u.cuda(), keep_logits=True)
    assert tuple(out.shape) == (B, 1 + steps)
    assert torch.equal(out[:, 0].cpu(), t["tensor"].reshape(B, -1)[:, n_prompt].long())      # starts from the first target token
    got = [l.cpu() for l in smp.last_ar["logits"]]
    # ---- sampling bit-exact from the kernel's own logits ----
    for i in range(steps):
        ids, _ = S.sample_tokens(got[i].numpy().copy(), 0.9, 40, 0.0, u[i].numpy())
        assert np.array_equal(out[:, i + 1].cpu().numpy(), ids), i
    # ---- logits vs the oracle's non-cached recompute of the same prefix ----
    num = O._Num(True)
    with torch.no_grad():
        n_enc = max(int(sum((~md[m.name]["input_mask"].reshape(B, -1)[b]).sum() for m in cfg.mods if m.in_enc)) for b in range(B))
        enc = O.select_encoder(P, cfg, md, n_enc, num)
        x = O.encoder_forward(P, cfg, enc["tokens"] + enc["emb"], enc["mask"], num)
        ctx = num.linear(x, P["decoder_proj_context.weight"], P["decoder_proj_context.bias"]) + enc["emb"]
        _, e, _ = O.embed_decoder_modality(P, spec, md[target])
        y_emb = e.float()[:, n_prompt:n_prompt + n_gen]
        table = P[f"decoder_embeddings.{target}.token_emb.weight"]
        worst = 0.0
        for i in range(steps):
            cur = i + 1
            prefix = out[:, :cur].cpu()
            y = table[prefix] + y_emb[:, :cur]
            causal = torch.ones(cur, cur, dtype=torch.bool).triu(1)[None].expand(B, -1, -1)
            yd = O.decoder_forward(P, cfg, y, ctx, enc["mask"], causal, num)
            want = num.linear(yd[:, -1], P[f"decoder_embeddings.{target}.to_logits.weight"], None)
            worst = max(worst, float((got[i] - want).norm() / want.norm()))
    record("generate.autoregressive_logits", case=case_name, worst_rel=worst, steps=steps)
    assert worst < 1.2e-2, worst
    # ---- end-of-sequence: decoding stops as soon as every sample has produced it (here: a batch of one, same draws) ----
    eos = int(out[0, 3])
    one = {k: {a: b[:1].cuda() for a, b in v.items()} for k, v in md.items()}
    out2 = smp.autoregressive_generate(one, target, temperature=0.9, top_k=40, top_p=0.0, use_eos=True, eos_token=eos, uniforms=u[:, :1].cuda())
    first = int((out[0] == eos).nonzero()[0])
    assert out2.shape[1] == first + 1 and torch.equal(out2[0], out[0, :first + 1])
    # ---- the upstream-shaped step: decode + merge back into mod_dict (tokenizer-level merge tested on the CPU) ----
    class Tok:
        vocab = {"[PAD]": 0, **{f"[S_{i}]": 5 + i for i in range(6)}}

        def get_vocab(self):
            return dict(self.vocab)

        def token_to_id(self, t):
            return self.vocab.get(t)
    dev_md = {k: {a: b.cuda() for a, b in v.items()} for k, v in md.items()}
    merged = smp.autoregressive_step_batched(dev_md, target, 0.9, 40, 0.0, use_eos=False, text_tokenizer=Tok(), seed=3)[target]
    assert merged["tensor"].shape[0] == B and merged["tensor"].shape == merged["input_mask"].shape == merged["target_mask"].shape
    assert merged["tensor"].is_cuda and torch.equal(merged["input_mask"], merged["target_mask"]) and not bool(merged["input_mask"][:, 0].any())
    # ---- classifier-free guidance: a second decoder state on the emptied conditioning, fp32 combination of the last-token logits ----
    cond_mod = next(m.name for m in cfg.mods if m.in_enc and m.name != target and m.kind == "tok" and bool((~md[m.name]["input_mask"]).any()))
    fresh = lambda: {k: {a: b.cuda() for a, b in v.items()} for k, v in md.items()}
    smp.autoregressive_generate(fresh(), target, 0.9, 40, 0.0, use_eos=False, uniforms=u.cuda(), keep_logits=True)
    lc0 = smp.last_ar["logits"][0].cpu().numpy()
    smp.autoregressive_generate(smp.unconditional_dict(fresh(), [cond_mod]), target, 0.9, 40, 0.0, use_eos=False, uniforms=u.cuda(), keep_logits=True)
    lu0 = smp.last_ar["logits"][0].cpu().numpy()
    assert float(np.abs(lc0 - lu0).max()) > 1e-3
    outg = smp.autoregressive_generate(fresh(), target, 0.9, 40, 0.0, use_eos=False, uniforms=u.cuda(), keep_logits=True,
                                       conditioning=[cond_mod], guidance_scale=3.0)
    assert np.array_equal(smp.last_ar["logits"][0].cpu().numpy(), S.cfg_logits(lc0, lu0, 3.0))        # same start token: exact
    for i in range(steps):                                                                             # sampling from the guided logits
        ids, _ = S.sample_tokens(smp.last_ar["logits"][i].cpu().numpy().copy(), 0.9, 40, 0.0, u[i].numpy())
        assert np.array_equal(outg[:, i + 1].cpu().numpy(), ids), i
    merged = smp.guided_autoregressive_step_batched(fresh(), target, 0.9, 40, 0.0, use_eos=False, text_tokenizer=Tok(), conditioning=[cond_mod],
                                                    guidance_scale=3.0, seed=3)[target]
    assert merged["tensor"].shape[0] == B and torch.equal(merged["input_mask"], merged["target_mask"])
    # ---- hipGraph replay of the per-position launch sequences: same logits bit for bit, same tokens; a second call reuses the graphs ----
    for rep in range(2):
        outr = smp.autoregressive_generate(fresh(), target, 0.9, 40, 0.0, use_eos=False, uniforms=u.cuda(), keep_logits=True, use_graphs=True)
        assert torch.equal(outr, out), rep
        for i in range(steps):
            assert torch.equal(smp.last_ar["logits"][i].cpu(), got[i]), (rep, i)
    assert len(smp._ar_graphs) == steps
    outg2 = smp.autoregressive_generate(fresh(), target, 0.9, 40, 0.0, use_eos=False, uniforms=u.cuda(), conditioning=[cond_mod], guidance_scale=3.0,
                                        use_graphs=True)
    assert torch.equal(outg2, outg)


class _Tok:
    """Stand-in for tokenizers.Tokenizer: the vocabulary lookups generation needs ([S_k], [EOS], [PAD])."""

    def __init__(self, n_sent=20, base=4):
        self.vocab = {"[PAD]": 0, "[EOS]": 3, **{f"[S_{k}]": base + k for k in range(n_sent)}}

    def get_vocab(self):
        return dict(self.vocab)

    def token_to_id(self, t):
        return self.vocab.get(t)


def test_generate_iter_and_sam_dense():
    """generate_iter yields after every schedule step and ends where generate ends (same seeds); guided MaskGIT steps expose every
    current prediction (write_all_predictions); generate_sam_dense merges the per-copy sequences into one (generate.py:1099-1161, :1230-1272)."""
    case = build_case("micro_swiglu")
    cfg = case["cfg"]
    model = build_hip_model(cfg, case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(case["sd"], strict=True)
    model = model.cuda().eval()
    target = "tok_a@32"
    spec = cfg.mod(target)
    md = gen_mod_dict(cfg, 1, target)
    dev_md = {k: {a: b.cuda() for a, b in v.items()} for k, v in md.items()}
    smp = sampler(model)
    n = spec.n_pos
    sched = [dict(target_domain=target, scheme="maskgit", num_tokens=k, temperature=1.0) for k in (n // 4, n // 4, n - 2 * (n // 4))]
    final = smp.generate(dev_md, sched, top_k=0, top_p=0.9, seed=5)
    seen = []
    for i, cur in enumerate(smp.generate_iter(dev_md, sched, top_k=0, top_p=0.9, seed=5)):
        seen.append(int((~cur[target]["input_mask"]).sum()))
    assert seen == [n // 4, 2 * (n // 4), n] and torch.equal(cur[target]["tensor"], final[target]["tensor"])
    assert bool(dev_md[target]["input_mask"].all())                                  # the caller's dict is untouched
    # guided step in iterator mode: all positions carry a prediction, the masks commit only num_tokens of them
    cond = [m.name for m in cfg.mods if m.in_enc and m.name != target][:1]
    gs = [dict(target_domain=target, scheme="maskgit", num_tokens=3, temperature=1.0, cfg_scale=2.0, cfg_cond_domains=cond)]
    plain = smp.generate(dev_md, gs, seed=9)
    shown = next(iter(smp.generate_iter(dev_md, gs, seed=9)))
    assert torch.equal(shown[target]["input_mask"], plain[target]["input_mask"]) and int((~shown[target]["input_mask"]).sum()) == 3
    committed = ~plain[target]["input_mask"].reshape(1, -1)
    a, b = shown[target]["tensor"].reshape(1, -1), plain[target]["tensor"].reshape(1, -1)
    assert torch.equal(a[committed], b[committed])
    assert a.shape[1] == n and torch.equal(a.long(), smp.last_step["samples"].long().reshape(1, -1))      # every position shows its sample
    # dense sequence prediction: 4 copies of one sample, one merged sequence back
    seq = next(m.name for m in cfg.mods if m.kind == "seq" and m.in_dec)
    tok = _Tok()
    md2 = O.synthetic_mod_dict(cfg, 1, 20, 0, seed=3, no_target=tuple(m.name for m in cfg.mods))
    for name, d in md2.items():
        d["target_mask"][:] = True
    t = md2[seq]
    L_ = t["tensor"].reshape(1, -1).shape[1]
    t["tensor"] = torch.randint(30, cfg.mod(seq).vocab, t["tensor"].shape, dtype=t["tensor"].dtype)
    t["tensor"].reshape(1, -1)[0, 2] = tok.vocab["[S_1]"]                           # input: two tokens, one sentinel
    t["input_mask"][:] = True; t["input_mask"].reshape(1, -1)[0, :3] = False
    t["tensor"].reshape(1, -1)[0, 3] = tok.vocab["[S_1]"]                           # the decoder continues from the sentinel
    t["target_mask"][:] = True; t["target_mask"].reshape(1, -1)[0, 3:min(L_, 9)] = False
    dev2 = {k: {a_: b_.cuda() for a_, b_ in v.items()} for k, v in md2.items()}
    s2 = [dict(target_domain=seq, temperature=1.0), dict(target_domain=target, scheme="maskgit", num_tokens=2, temperature=1.0)]
    out = smp.generate_sam_dense(dev2, s2, tok, batch_size=4, key=seq, top_k=50, seed=1)
    o = out[seq]
    assert o["tensor"].shape[0] == 1 and o["tensor"].dim() == 2 and not bool(o["input_mask"].any()) and bool(o["target_mask"].all())
    ref = smp.generate({k: {a_: (b_.expand(4, *b_.shape[1:]).contiguous() if torch.is_tensor(b_) else b_) for a_, b_ in v.items()} for k, v in dev2.items()},
                       s2[:1], text_tokenizer=tok, top_k=50, seed=1)
    sent = smp.sentinel_ids(tok)
    want = []
    for i in range(4):
        r = ref[seq]
        want += smp.merge_span_masking(r["tensor"][i][r["input_mask"][i] == 0].tolist(), r["tensor"][i][r["target_mask"][i] == 0].tolist(), sent)
    assert o["tensor"][0].tolist() == want and len(want) >= 4 * 2
    assert torch.equal(out[target]["tensor"], dev2[target]["tensor"])               # other modalities pass through
