"""End-to-end parity of the HIP model on a real MI355X:
  * token selection / masks / ids: bit-exact against the upstream fixtures;
  * loss, per-modality losses, logits, gradients: against the CPU oracle run with bf16 rounding at
    upstream's autocast points (tight) and against the upstream fp32 fixture (bf16-level tolerance).
Tolerances are stated at each assert."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import fourm_oracle as O
from tests.golden.cases import build_case
from tests.parity_log import record
from tests.util_model import build_hip_model, tie, to_device

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
HIP_CASES = ["micro_swiglu", "micro_pad", "micro_gelu", "micro_qknorm", "ti_mod7", "l_like", "b_mod7", "l_mod21"]


def setup(name):
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    case = build_case(name)
    model = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(case["sd"], strict=True)
    model = model.cuda().train()
    return g, case, model


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


@pytest.mark.parametrize("name", HIP_CASES)
def test_selection_bit_exact(name):
    g, case, model = setup(name)
    md = to_device(case["mod_dict"])
    tok, emb, mask, mod = model.forward_mask_encoder(md, case["N"])
    assert np.array_equal(mask.cpu().numpy(), g["enc/mask"])
    assert np.array_equal(mod.cpu().numpy(), g["enc/mod_mask"])
    random.seed(case["order_seed"])
    dtok, demb, dmask, tgt, attn, dmod = model.forward_mask_decoder(md, case["M"])
    assert np.array_equal(dmask.cpu().numpy(), g["dec/mask"])
    assert np.array_equal(dmod.cpu().numpy(), g["dec/mod_mask"])
    assert np.array_equal(tgt.cpu().numpy(), g["dec/target_ids"])
    assert np.array_equal(np.packbits(attn.cpu().numpy(), axis=-1), g["dec/attn_mask"])
    if "enc/emb" in g:
        # gathered rows are copies of table rows: exact.  Rows produced by the pixel / T5 projections
        # come from a bf16 GEMM: 2^-8 relative.
        assert np.array_equal(emb.cpu().numpy(), g["enc/emb"])
        assert np.array_equal(demb.cpu().numpy(), g["dec/emb"])
        assert np.array_equal(dtok.cpu().numpy(), g["dec/tokens"])
        ref = torch.from_numpy(g["enc/tokens"])
        dense_ids = [m.id for m in case["cfg"].mods if m.kind in ("patch", "seq_emb")]
        dense = torch.zeros_like(mod.cpu(), dtype=torch.bool)
        for i in dense_ids:
            dense |= mod.cpu() == i
        assert torch.equal(tok.cpu()[~dense], ref[~dense])
        if dense.any():
            assert rel(tok.cpu()[dense], ref[dense]) < 8e-3
    else:
        np.testing.assert_allclose(emb.sum(-1).cpu().numpy(), g["enc/emb_rowsum"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(demb.sum(-1).cpu().numpy(), g["dec/emb_rowsum"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", HIP_CASES)
def test_loss_and_gradients(name):
    g, case, model = setup(name)
    cfg, md = case["cfg"], case["mod_dict"]
    order = g["meta/order"].tolist()
    # oracle with bf16 rounding at the autocast points, same weights, CPU autograd
    P = tie({k: v.clone().requires_grad_(v.is_floating_point()) for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    o_loss, o_mod = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"], emulate_bf16=True)
    o_loss.sum().backward()
    random.seed(case["order_seed"])
    loss, mod_loss = model(to_device(md), case["N"], case["M"], loss_type=case["loss_type"])
    loss.backward()
    torch.cuda.synchronize()
    # loss: bf16 pipeline vs bf16-emulating oracle 3e-3, vs upstream fp32 2e-2
    assert abs(float(loss) - float(o_loss.sum())) < 3e-3 * abs(float(o_loss.sum())), (float(loss), float(o_loss.sum()))
    assert abs(float(loss) - float(g["loss"][0])) < 2e-2 * abs(float(g["loss"][0]))
    for k, v in mod_loss.items():
        assert abs(float(v) - float(o_mod[k].sum())) < 5e-3 * max(1.0, abs(float(o_mod[k].sum()))), k
    # gradients: relative Frobenius error per tensor vs the bf16-emulating oracle
    worst = []
    for n, p in model.named_parameters():
        og = P[n].grad
        if og is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            continue
        assert p.grad is not None, n
        if float(og.norm()) < 1e-9:
            assert float(p.grad.norm()) < 1e-6, n
            continue
        worst.append((rel(p.grad, og), n))
    worst.sort(reverse=True)
    record("model.loss_and_gradients", case=name, loss_hip=float(loss), loss_bf16_oracle=float(o_loss.sum()), loss_fp32_upstream=float(g["loss"][0]),
           grad_rel_worst=worst[0][0], grad_rel_worst_name=worst[0][1], grad_rel_median=float(np.median([w[0] for w in worst])))
    # measured r02: worst tensor 1.0e-2 ... 2.4e-2 (a LayerNorm weight of a deep decoder block), median 5.6e-3 ... 1.0e-2
    assert worst[0][0] < 4.8e-2, worst[:8]
    assert float(np.median([w[0] for w in worst])) < 2e-2, worst[:8]


# Measured on MI355X (gpurun_out/parity.jsonl -> profiles/r02_parity.jsonl): per-modality relative Frobenius error of
# the bf16 HIP logits against (a) the oracle with bf16 rounding at upstream's autocast points, (b) the upstream fp32
# fixture.  The asserts hold them to <= 2x the measured worst case per model (never looser than that).
#   measured r02 (worst modality)   HIP vs bf16 oracle   HIP vs fp32   [bf16 oracle vs fp32 = upstream's own autocast gap]
#   micro_swiglu                    7.2e-3               7.6e-3        7.6e-3
#   ti_mod7  (4M-Ti, 6+6)           5.6e-3               7.4e-3        7.5e-3
#   b_mod7   (4M-B, 12+12, benched) 5.9e-3               9.7e-3        9.6e-3
LOGIT_BOUNDS = {"micro_swiglu": (1.4e-2, 1.5e-2), "ti_mod7": (1.1e-2, 1.5e-2), "b_mod7": (1.2e-2, 1.9e-2), "l_mod21": (1.2e-2, 1.9e-2)}
#   l_mod21  (4M-L, 24+24)          5.9e-3               9.6e-3        9.6e-3      (r03; r04 with the context-norm hoist: see profiles/r04_parity.jsonl)


@pytest.mark.parametrize("name", ["micro_swiglu", "ti_mod7", "b_mod7", "l_mod21"])
def test_logits(name):
    g, case, model = setup(name)
    cfg = case["cfg"]
    model.eval()
    random.seed(case["order_seed"])
    with torch.no_grad():
        logits = model(to_device(case["mod_dict"]), case["N"], case["M"], return_logits=True)
    order = g["meta/order"].tolist()
    P = tie({k: v.clone() for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    with torch.no_grad():
        emu = O.fourm_forward(P, cfg, case["mod_dict"], case["N"], case["M"], order, return_logits=True, emulate_bf16=True)
        f32 = O.fourm_forward(P, cfg, case["mod_dict"], case["N"], case["M"], order, return_logits=True)
    b_emu, b_f32 = LOGIT_BOUNDS[name]
    errs = {}
    for k, v in logits.items():
        assert tuple(v.shape) == tuple(f32[k].shape) == (case["mod_dict"][k]["tensor"].shape[0], case["M"], cfg.mod(k).vocab)
        # the oracle's fp32 logits are the upstream model's (pinned by make_golden.py; norm + head slice rechecked here)
        assert abs(float(f32[k].double().norm()) - float(g[f"logits_fro/{k}"])) < 1e-4 * float(g[f"logits_fro/{k}"]), k
        errs[k] = dict(hip_vs_bf16_oracle=rel(v, emu[k]), hip_vs_fp32=rel(v, f32[k]), bf16_oracle_vs_fp32=rel(emu[k], f32[k]))
    record("model.logits", case=name, per_modality=errs)
    worst_emu = max(e["hip_vs_bf16_oracle"] for e in errs.values())
    worst_f32 = max(e["hip_vs_fp32"] for e in errs.values())
    assert worst_emu < b_emu, errs
    assert worst_f32 < b_f32, errs


@pytest.mark.parametrize("name", ["micro_swiglu", "micro_pad", "micro_gelu", "micro_qknorm", "ti_mod7", "l_like", "b_mod7"])
def test_fp32_verification_mode(name):
    """compute_precision = "fp32": the SAME engine (selection, launch sequence, hand-written backward, segmented heads) on the
    fp32 verification kernels (csrc/fp32_verify.hip), no bf16 rounding anywhere, against the upstream fp32 model (the oracle in
    fp32 == upstream, pinned by make_golden.py; the fixture's loss / logit norms / gradient norms are re-checked here).
    north_star asks for logits within 1e-3 relative.  Measured r02 (profiles/r02_parity.jsonl): logits 4.8e-7 ... 8.1e-7, worst
    gradient tensor 1.1e-6 ... 4.7e-6, loss <= 2e-7; held to 1e-5 / 5e-5 / 2e-6 (fp32 summation-order noise only)."""
    g, case, model = setup(name)
    model.compute_precision = "fp32"
    cfg, md = case["cfg"], case["mod_dict"]
    order = g["meta/order"].tolist()
    P = tie({k: v.clone().requires_grad_(v.is_floating_point()) for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    o_loss, o_mod = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"])
    o_loss.sum().backward()
    random.seed(case["order_seed"])
    loss, mod_loss = model(to_device(md), case["N"], case["M"], loss_type=case["loss_type"])
    loss.backward()
    torch.cuda.synchronize()
    assert model.engine.fp32 and model.engine.adt == torch.float32
    e_loss = abs(float(loss) - float(g["loss"][0])) / abs(float(g["loss"][0]))          # vs the UPSTREAM fixture
    for k, v in mod_loss.items():
        assert abs(float(v) - float(g[f"mod_loss/{k}"][0])) < 2e-5 * max(1.0, abs(float(g[f"mod_loss/{k}"][0]))), k
    worst = []
    for n, p in model.named_parameters():
        og = P[n].grad
        if og is None or float(og.norm()) < 1e-9:
            assert p.grad is None or float(p.grad.norm()) < 1e-6, n
            continue
        worst.append((rel(p.grad, og), n))
        key = f"grad_l2/{n}"
        if key in g.files:                                                               # upstream's own gradient norm
            assert abs(float(p.grad.double().norm()) - float(g[key])) < 1e-3 * float(g[key]) + 1e-7, n
    worst.sort(reverse=True)
    # logits
    model.eval()
    random.seed(case["order_seed"])
    with torch.no_grad():
        logits = model(to_device(md), case["N"], case["M"], return_logits=True)
        f32 = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, return_logits=True)
    e_logits = {k: rel(v, f32[k]) for k, v in logits.items()}
    for k, v in logits.items():
        assert v.dtype == torch.float32
        assert abs(float(v.double().norm()) - float(g[f"logits_fro/{k}"])) < 1e-4 * float(g[f"logits_fro/{k}"]), k
    record("model.fp32_mode", case=name, loss_rel=e_loss, logits_rel_worst=max(e_logits.values()), grad_rel_worst=worst[0][0],
           grad_rel_worst_name=worst[0][1], grad_rel_median=float(np.median([w[0] for w in worst])))
    assert e_loss < 2e-6, e_loss
    assert max(e_logits.values()) < 1e-5, e_logits
    assert worst[0][0] < 5e-5, worst[:6]


def test_eval_forward_and_accumulation():
    g, case, model = setup("micro_swiglu")
    md = to_device(case["mod_dict"])
    random.seed(1); l1, _ = model(md, case["N"], case["M"]); l1.backward()
    g1 = {n: p.grad.clone() for n, p in model.named_parameters()}
    random.seed(1); l2, _ = model(md, case["N"], case["M"]); l2.backward()      # accumulates (no zero_grad between)
    for n, p in model.named_parameters():
        assert rel(p.grad, 2 * g1[n]) < 1e-3 or float(g1[n].norm()) == 0, n
    for p in model.parameters():
        p.grad = None
    random.seed(1); l3, _ = model(md, case["N"], case["M"]); (0.5 * l3).backward()
    for n, p in model.named_parameters():
        assert rel(p.grad, 0.5 * g1[n]) < 2e-2 or float(g1[n].norm()) == 0, n
    model.eval()
    with torch.no_grad():
        random.seed(1); l4, ml = model(md, case["N"], case["M"])
    assert abs(float(l4) - float(l1)) < 1e-6 and not l4.requires_grad


@pytest.mark.parametrize("name", ["micro_swiglu", "micro_gelu", "ti_mod7"])
def test_embedding_and_cat_sub_api(name):
    """The pieces generation code calls directly: every embedding's own forward() / forward_embed() and
    cat_encoder_tensors / cat_decoder_tensors (fm.py:245-336) against the oracle's per-modality embedders."""
    g, case, model = setup(name)
    cfg, P = case["cfg"], tie(dict(case["sd"]), case["cfg"], case["share_embedding"])
    md = to_device(case["mod_dict"])
    num = O._Num(False)
    enc_x, enc_e, dec = {}, {}, {}
    with torch.no_grad():
        for spec in cfg.mods:
            d_cpu = case["mod_dict"][spec.name]
            if spec.in_enc:
                x, e = O.embed_encoder_modality(P, spec, d_cpu, num)
                enc_x[spec.name], enc_e[spec.name] = x.float(), e
                out = model.encoder_embeddings[spec.name](dict(md[spec.name]))
                tol = 2e-2 if spec.kind in ("patch", "seq_emb") else 0.0          # bf16 projection vs gathers
                assert rel(out["x"], x) <= tol, (spec.name, "x", rel(out["x"], x))
                assert torch.equal(out["emb"].cpu(), e.float()), (spec.name, "emb")
            if spec.in_dec:
                x, e, ids = O.embed_decoder_modality(P, spec, d_cpu)
                dec[spec.name] = (x, e, ids)
                out = model.decoder_embeddings[spec.name].forward_embed(dict(md[spec.name]))
                assert torch.equal(out["x"].cpu(), x) and torch.equal(out["emb"].cpu(), e.float()), spec.name
                assert torch.equal(out["ids"].cpu().reshape(ids.shape).long(), ids.long())
        # encoder concatenation, in mod_dict order
        names = [n for n in case["mod_dict"] if n in enc_x]
        tok, emb, mask, mod = model.cat_encoder_tensors(md)
        want_mask = torch.cat([case["mod_dict"][n]["input_mask"].bool().reshape(tok.shape[0], -1) for n in names], 1)
        assert torch.equal(mask.cpu(), want_mask)
        assert torch.equal(emb.cpu(), torch.cat([enc_e[n].float() for n in names], 1))
        assert rel(tok, torch.cat([enc_x[n] for n in names], 1)) < 2e-2
        want_mod = torch.cat([torch.full(enc_e[n].shape[:2], cfg.mod(n).id, dtype=torch.int16) for n in names], 1)
        assert torch.equal(mod.cpu(), want_mod)
        # decoder concatenation: shuffled order, teacher forcing on sequences, mask token on grids
        random.seed(case["order_seed"])
        tok, emb, mask, tgt, dam, mod = model.cat_decoder_tensors(md)
        order = [str(n) for n in g["meta/order"]]
        xs, es, ms, ts, am = [], [], [], [], []
        for n in order:
            x, e, ids = dec[n]
            d_cpu = case["mod_dict"][n]
            tm = d_cpu["target_mask"].bool().reshape(x.shape[0], -1)
            a = d_cpu["decoder_attention_mask"].reshape(x.shape[0], -1)
            if cfg.mod(n).is_seq:
                xs.append(x[:, :-1]); es.append(e[:, :-1]); ts.append(ids[:, 1:]); ms.append(tm[:, 1:] | tm[:, :-1]); am.append(a[:, :-1])
            else:
                xs.append(P["mask_token"].expand(x.shape[0], x.shape[1], -1)); es.append(e); ts.append(ids); ms.append(tm); am.append(a)
        assert torch.equal(tok.cpu(), torch.cat(xs, 1)) and torch.equal(emb.cpu(), torch.cat(es, 1).float())
        assert torch.equal(mask.cpu(), torch.cat(ms, 1)) and torch.equal(tgt.cpu(), torch.cat(ts, 1).long())
        assert torch.equal(dam.cpu().int(), torch.cat(am, 1).int())


def test_unsupported_config_is_loud():
    """head_dim != 64 has no attention kernel: the model must refuse, not fall back."""
    import dataclasses
    case = build_case("micro_swiglu")
    cfg = dataclasses.replace(case["cfg"], heads=4)          # dim 128 / 4 heads = head_dim 32
    model = build_hip_model(cfg).cuda()
    with pytest.raises(NotImplementedError):
        model(to_device(case["mod_dict"]), case["N"], case["M"])


def test_fused_adamw_step_matches_torch():
    from fourm.utils.optim_factory import FusedAdamW
    g, case, model = setup("micro_swiglu")
    md = to_device(case["mod_dict"])
    ref = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    ref.load_state_dict(case["sd"]); ref = ref.cuda()
    opt = FusedAdamW([{"params": [p for n, p in model.named_parameters() if p.dim() > 1], "weight_decay": 0.05},
                      {"params": [p for n, p in model.named_parameters() if p.dim() <= 1], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    ropt = torch.optim.AdamW([{"params": [p for n, p in ref.named_parameters() if p.dim() > 1], "weight_decay": 0.05},
                              {"params": [p for n, p in ref.named_parameters() if p.dim() <= 1], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    for step in range(2):
        random.seed(step); loss, _ = model(md, case["N"], case["M"]); loss.backward()
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            q.grad = p.grad.clone()
        norm = opt.fused_grad_norm(clip=1.0)
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
        opt.step(); ropt.step()
        opt.zero_grad(); ropt.zero_grad()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert float((p - q).abs().max()) < 2e-6, n
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 2.0
    # the lazy norm (no clipping: sum g^2 rides on the AdamW launches, filled by step()) equals the eager one and leaves the same update
    random.seed(5); loss, _ = model(md, case["N"], case["M"]); loss.backward()
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        q.grad = p.grad.clone()
    eager = float(opt.fused_grad_norm())
    lazy = opt.fused_grad_norm(lazy=True)
    assert float(lazy) == 0.0                                   # not computed yet ...
    opt.step(); ropt.step()
    want = float(torch.nn.utils.get_total_norm([q.grad for q in ref.parameters() if q.grad is not None])) if hasattr(torch.nn.utils, "get_total_norm") else eager
    assert abs(float(lazy) - eager) < 1e-5 * eager and abs(eager - want) < 1e-5 * want, (float(lazy), eager, want)      # ... filled by the step
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert float((p - q).abs().max()) < 2e-6, n


def test_hoisted_context_norm_with_frozen_parts():
    """The context-norm hoist (engine.hoist_ctx) computes dL/d(W diag(gamma)) once per decoder block and unfolds it into dW and dgamma:
    freezing either tensor alone must leave the other one's gradient unchanged (and FOURM_HOIST_CTX=0, the per-block form, agrees)."""
    g, case, model = setup("micro_swiglu")
    assert model.engine.hoist_ctx
    md = to_device(case["mod_dict"])

    def grads(freeze):
        _, _, m = setup("micro_swiglu")
        for blk in m.decoder:
            if freeze == "kv":
                blk.cross_attn.kv.weight.requires_grad_(False)
            if freeze == "norm":
                blk.context_norm.weight.requires_grad_(False)
        random.seed(case["order_seed"])
        loss, _ = m(md, case["N"], case["M"])
        loss.backward()
        return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, float(loss)
    full, l0 = grads(None)
    for freeze, kept, gone in (("kv", "context_norm.weight", "cross_attn.kv.weight"), ("norm", "cross_attn.kv.weight", "context_norm.weight")):
        part, l1 = grads(freeze)
        assert l1 == l0
        assert not any(gone in n and "decoder." in n for n in part), freeze
        for n, gr in full.items():
            if kept in n and "decoder." in n:
                assert rel(part[n], gr) < 2e-5, (freeze, n, rel(part[n], gr))
        for n in ("decoder.0.self_attn.qkv.weight", "encoder.0.attn.qkv.weight"):            # ... and nothing else moves
            assert rel(part[n], full[n]) < 2e-5, (freeze, n)


def test_trainer_step_trajectory_b_mod7():
    """The trainer's update - NativeScalerWithGradNormCount(loss, FusedAdamW, clip_grad=...) as run_training_4m.py:727-733 calls it - against
    torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over THREE steps of 4M-B with a changing learning rate: the weight trajectory, not
    only the first step.  The torch side receives the HIP gradients of every step (copied between the backward and the update: the
    fp32 atomics of the weight-gradient GEMMs are not bit-reproducible, and Adam turns a sign flip of a noise-level gradient into a full
    +-lr step), so the two trajectories can only separate through the clip / AdamW arithmetic itself."""
    from fourm.utils.native_scaler import NativeScalerWithGradNormCount
    from fourm.utils.optim_factory import FusedAdamW
    g, case, model = setup("b_mod7")
    md = to_device(case["mod_dict"])
    ref = {n: p.detach().clone().requires_grad_(True) for n, p in model.named_parameters()}
    groups = lambda named: [{"params": [p for n, p in named if p.dim() > 1], "weight_decay": 0.05},
                            {"params": [p for n, p in named if p.dim() <= 1], "weight_decay": 0.0}]
    opt = FusedAdamW(groups(list(model.named_parameters())), lr=1e-3, betas=(0.9, 0.95))
    ropt = torch.optim.AdamW(groups(list(ref.items())), lr=1e-3, betas=(0.9, 0.95))
    fused_norm = opt.fused_grad_norm

    def norm_and_hand_over(clip=None, lazy=False):  # called by the scaler between backward and step: the same gradients go to the torch side
        for n, p in model.named_parameters():
            ref[n].grad = None if p.grad is None else p.grad.clone()
        return fused_norm(clip=clip, lazy=lazy)
    opt.fused_grad_norm = norm_and_hand_over
    scaler = NativeScalerWithGradNormCount(enabled=False)
    clip = 0.5                                   # below the gradient norm of the seeded model: the clip is active on every step
    norms, losses = [], []
    for step, lr in enumerate((1e-3, 7e-4, 3e-4)):
        for o in (opt, ropt):
            for gr in o.param_groups:
                gr["lr"] = lr
        random.seed(step); loss, _ = model(md, case["N"], case["M"])
        norm = scaler(loss, opt, clip_grad=clip, parameters=model.parameters())
        opt.zero_grad()
        rnorm = torch.nn.utils.clip_grad_norm_([p for p in ref.values() if p.grad is not None], clip)
        ropt.step(); ropt.zero_grad()
        norms.append((float(norm), float(rnorm))); losses.append(float(loss))
        assert abs(float(norm) - float(rnorm)) < 1e-4 * float(rnorm) and float(rnorm) > clip, norms
    worst = max((float((p - ref[n]).abs().max()), n) for n, p in model.named_parameters())
    moved = max(float((p.detach().cpu() - case["sd"][n]).abs().max()) for n, p in model.named_parameters() if n in case["sd"])
    record("model.trainer_trajectory", case="b_mod7", steps=3, worst_abs_diff=worst[0], worst_name=worst[1], largest_move=moved, norms=norms, losses=losses)
    assert moved > 1e-3 and worst[0] < 5e-6 and losses[2] < losses[0], (worst, moved, losses)


def test_fused_adamw_rewrites_weight_shadows():
    """FusedAdamW.step() updates the weight matrices and rewrites their plain bf16 GEMM-operand copies in the same streaming
    kernel (fm_adamw_shadow): after a step those shadows are marked current AND equal the cast of their fp32 master; the
    transposed copies (dX operands) are left stale for the engine's single refresh launch at the next forward."""
    from fourm.utils.optim_factory import FusedAdamW
    g, case, model = setup("micro_swiglu")
    md = to_device(case["mod_dict"])
    opt = FusedAdamW([{"params": [p for n, p in model.named_parameters() if p.dim() > 1], "weight_decay": 0.05},
                      {"params": [p for n, p in model.named_parameters() if p.dim() <= 1], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.95))
    eng = model.engine

    def check(only_plain):
        n = 0
        for key, sh in eng.shadows.items():
            plain = all(not j[2] and len(j) == 3 for j in sh.jobs)     # (images with a folded LayerNorm weight depend on two parameters: lazy too)
            if only_plain and not plain:
                assert sh.stamp != eng._stamp(sh.params), key          # stale: refreshed lazily, never used as is
                continue
            for j in sh.jobs:
                p, dst, transposed = j[:3]
                want = p.detach().reshape(p.shape[0], -1)
                if len(j) > 3:
                    want = want * j[3].detach()[None, :]                # W diag(gamma): the context-norm hoist's kv image
                want = want.to(torch.bfloat16)
                want = want.t() if transposed else want
                assert torch.equal(dst[: want.shape[0], : want.shape[1]], want), (key, transposed)
            assert sh.stamp == eng._stamp(sh.params), key
            n += 1
        return n
    for step in range(2):
        random.seed(step); loss, _ = model(md, case["N"], case["M"]); loss.backward()
        assert check(only_plain=False) > 30                            # after a forward every copy is current
        before = {k: s.buf.clone() for k, s in eng.shadows.items()}
        opt.step(); opt.zero_grad()
        torch.cuda.synchronize()
        assert check(only_plain=True) > 15
        assert sum(int(not torch.equal(before[k], s.buf)) for k, s in eng.shadows.items()) > 15      # the weights did move
    random.seed(5); l2, _ = model(md, case["N"], case["M"])
    assert check(only_plain=False) > 30
    assert float(l2) < float(loss)                                      # and the forward sees them


def test_graphed_train_step_matches_eager():
    """GraphedTrainStep (fourm/hip/graph.py): the whole step captured in one hipGraph and replayed on new batches with a changing
    learning rate follows the eager launch sequence: same losses, same weights (up to the summation order of the fp32 atomics in the
    weight-gradient kernels, which differs run to run in the eager path as well)."""
    from fourm.hip.graph import GraphedTrainStep
    from fourm.utils.optim_factory import FusedAdamW
    case = build_case("micro_swiglu")

    def make():
        m = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
        m.load_state_dict(case["sd"]); m = m.cuda().train()
        o = FusedAdamW([{"params": [p for n, p in m.named_parameters() if p.dim() > 1], "weight_decay": 0.05},
                        {"params": [p for n, p in m.named_parameters() if p.dim() <= 1], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
        return m, o
    batches = [to_device(O.synthetic_mod_dict(case["cfg"], 3, 20, 18, seed=50 + i)) for i in range(5)]
    lrs = [1e-3 * (1 + 0.3 * i) for i in range(5)]
    # eager reference, run TWICE: the decoder order is re-drawn from the same RNG state before every forward (the graph freezes one
    # order); the distance between the two identical eager runs is the noise floor of the fp32 atomics amplified by Adam
    def run_eager():
        m, o = make()
        out = []
        for i, b in enumerate(batches):
            for g in o.param_groups:
                g["lr"] = lrs[i]
            random.seed(11); loss, _ = m(b, case["N"], case["M"]); loss.backward()
            norm = o.fused_grad_norm(clip=1.0); o.step(); o.zero_grad(set_to_none=True)
            out.append((float(loss), float(norm)))
        return m, out
    me, eager = run_eager()
    me2, eager2 = run_eager()

    def dist(ma, mb):
        num = sum(float((p - q).double().pow(2).sum()) for p, q in zip(ma.parameters(), mb.parameters()))
        den = sum(float((p.double() - torch.zeros_like(p, dtype=torch.float64)).pow(2).sum()) for p in ma.parameters())
        return (num / den) ** 0.5
    floor = dist(me, me2)
    # graphed: warm-up steps of the constructor run on batch 0 with lr[0]; rebuild the starting point afterwards
    mg, og = make()
    gs = GraphedTrainStep(mg, og, batches[0], case["N"], case["M"], clip_grad=1.0, order_seed=11)
    mg.load_state_dict(case["sd"])                       # back to the initial weights (in place: the flat store does not move)
    for st in og.state.values():
        st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    gs.resync()                                          # weights changed outside the graph: rebuild the bf16 shadows
    graphed = []
    for i, b in enumerate(batches):
        for g in og.param_groups:
            g["lr"] = lrs[i]
        loss, _, norm = gs.step(b)
        graphed.append((float(loss), float(norm)))
    # (Adam's first steps move every weight by ~lr * sign(g): a gradient entry whose sign hangs on the summation order of the fp32
    # atomics lands 2 lr apart in two runs of the SAME eager code as well; hence a distribution check, not bit equality)
    # tolerance per step: 3e-4 / 1e-3 relative (measured between graph and eager: up to 1.04e-4 at the fourth step), or four times what
    # two runs of the SAME eager code differ by at that step, whichever is larger
    for (le, ne), (le2, ne2), (lg, ng) in zip(eager, eager2, graphed):
        assert abs(le - lg) < max(3e-4 * abs(le), 4 * abs(le - le2)), (eager, eager2, graphed)
        assert abs(ne - ng) < max(1e-3 * abs(ne), 4 * abs(ne - ne2)), (eager, eager2, graphed)
    assert abs(eager[0][0] - eager[-1][0]) > 1e-3          # the weights did move
    d_graph = dist(me, mg)
    record("graph.vs_eager", eager_vs_eager=floor, graph_vs_eager=d_graph, losses_eager=[e[0] for e in eager], losses_graph=[g_[0] for g_ in graphed])
    # a frozen learning rate or a stale shadow would put the weights ~6e-2 apart (30 % of lr per step on weights of std 0.02); the
    # atomics noise of two eager runs measured 2e-4 .. 6e-4 here, so the bound is a multiple of the measured floor with an absolute
    # allowance for the run where the two eager samples happen to agree closely
    assert d_graph < max(5 * floor, 2e-3), (d_graph, floor)
    assert all(float(st["step"]) == 5.0 for st in og.state.values())


# ------------------------------------------------------------------------------------------------------------------------------------
# where the bf16 error comes from: the residual stream after every block (VERDICT r02 weak #1 / next #4a)
# ------------------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["ti_mod7", "b_mod7"])
def test_residual_stream_error_follows_the_autocast_curve(name):
    """After EVERY encoder / decoder block the HIP (bf16) residual stream is compared with the oracle's - run with bf16 rounding at
    upstream's autocast points (``emu``) and in fp32.  e_hip(i) = |hip - emu| / |emu| must stay below the distance the two correct
    pipelines are allowed to have: a kernel with a systematic bias would lift the curve from its block on, while pure rounding noise
    grows like the emulating oracle's own distance from fp32, e_ref(i) = |emu - f32| / |f32| (both pipelines round the same operands
    to bf16, only the accumulation orders differ, so e_hip stays BELOW e_ref).  The end-to-end logit bound cannot see this."""
    g, case, model = setup(name)
    cfg, md = case["cfg"], case["mod_dict"]
    order = g["meta/order"].tolist()
    P = tie({k: v.clone() for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    t_emu, t_f32 = {}, {}
    with torch.no_grad():
        O.fourm_forward(P, cfg, md, case["N"], case["M"], order, emulate_bf16=True, taps=t_emu)
        O.fourm_forward(P, cfg, md, case["N"], case["M"], order, taps=t_f32)
    random.seed(case["order_seed"])
    loss, _ = model(to_device(md), case["N"], case["M"])
    torch.cuda.synchronize()
    c = model.engine._ctx
    st, B = c["st"], c["enc"]["B"]
    N, M, D = c["enc"]["Nt"], c["dec"]["Nt"], cfg.dim

    def stream(buf, n):
        return buf[: B * n].view(B, n, D).float().cpu()
    Le, Ld = len(st["enc_layers"]), len(st["dec_layers"])
    hip = {}
    for i in range(Le):
        hip[f"enc_block{i}"] = stream(st["enc_layers"][i + 1]["x_in"] if i + 1 < Le else st["x_final"], N)
    hip["context"] = stream(st["ctx"], N)
    for i in range(Ld):
        hip[f"dec_block{i}"] = stream(st["dec_layers"][i + 1]["y_in"] if i + 1 < Ld else st["y_final"], M)
    names = [f"enc_block{i}" for i in range(Le)] + ["context"] + [f"dec_block{i}" for i in range(Ld)]
    e_hip = [rel(hip[k], t_emu[k]) for k in names]
    e_ref = [rel(t_emu[k], t_f32[k]) for k in names]
    record("model.residual_stream_taps", case=name, taps=names, hip_vs_bf16_oracle=e_hip, bf16_oracle_vs_fp32=e_ref)
    loss.backward()          # (releases the saved state)
    for k, a, b in zip(names, e_hip, e_ref):
        # measured r03 (4M-B): e_hip 3.0e-3 (first encoder block) ... 5.8e-3 (last), 3.5e-3 ... 4.8e-3 over the decoder; e_ref 4.1e-3 ... 8.2e-3
        # and 5.7e-3 ... 8.8e-3; ratio 0.55 ... 0.75 at every tap, no step anywhere (profiles/r03_parity.jsonl)
        assert a < 0.85 * b + 2e-4, (k, a, b, list(zip(names, e_hip, e_ref)))
    # no block adds more than the whole curve's final level (a jump = a biased kernel entering at that block)
    jumps = [e_hip[i] - e_hip[i - 1] for i in range(1, len(e_hip))]
    assert max(jumps) < 0.5 * e_ref[-1] + 2e-4, (max(jumps), e_ref[-1], list(zip(names, e_hip)))


@pytest.mark.parametrize("name", ["micro_swiglu", "ti_mod7"])
def test_activation_checkpointing(name):
    """use_act_checkpoint=True (fm.py:103-113): only block inputs are kept, every block's activations are recomputed in the backward.
    Same loss bit for bit, the same gradients (up to the atomics' summation order), a smaller workspace."""
    res = {}
    for ckpt in (False, True):
        g, case, model = setup(name)
        model.use_act_checkpoint = ckpt
        random.seed(case["order_seed"])
        loss, _ = model(to_device(case["mod_dict"]), case["N"], case["M"], loss_type=case["loss_type"])
        loss.backward()
        torch.cuda.synchronize()
        eng = model._engine
        res[ckpt] = (float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}, eng.ws.nbytes(),
                     sum(1 for k in eng.ws.bufs if ".h1" in str(k)))
    assert res[True][0] == res[False][0]
    assert set(res[True][1]) == set(res[False][1])
    for n, gr in res[False][1].items():
        assert rel(res[True][1][n], gr) < 2e-5 or float(gr.norm()) < 1e-9, (n, rel(res[True][1][n], gr))
    # (r04: the context-norm hoist took the per-layer normalised context copies out of the plain workspace and added the all-layer d(kv)
    # buffer to both: 0.81 on the 2 + 2 block micro model, where the embeddings dominate)
    assert res[True][2] < 0.85 * res[False][2], (res[True][2], res[False][2])
    record("model.activation_checkpointing", case=name, workspace_bytes=res[False][2], workspace_bytes_checkpointed=res[True][2])


def _with_drop_path(model, cfg):
    from fourm.models.fm_utils import DropPath
    pe = [0.2 + 0.1 * i for i in range(cfg.enc_depth)]
    pd = [0.3 + 0.1 * i for i in range(cfg.dec_depth)]
    for blk, p in zip(list(model.encoder) + list(model.decoder), pe + pd):
        blk.drop_path = DropPath(p)
    return pe, pd


@pytest.mark.parametrize("name", ["micro_swiglu", "micro_gelu"])
def test_drop_path(name):
    """Stochastic depth (fm_utils.py:64-87): with the per-sample uniforms replayed, loss and gradients follow the oracle that applies the same
    scales (pinned to upstream in test_model_cpu.py); checkpointing recomputes with the SAME draws; eval mode ignores it."""
    g, case, model = setup(name)
    cfg, md = case["cfg"], case["mod_dict"]
    pe, pd = _with_drop_path(model, cfg)
    B = next(iter(md.values()))["tensor"].shape[0]
    gen = torch.Generator().manual_seed(4)
    draws = [torch.rand(B, generator=gen) for _ in range(2 * cfg.enc_depth + 3 * cfg.dec_depth)]
    drop = {"enc": [[O.drop_path_scale(draws[2 * i + j], pe[i]) for j in range(2)] for i in range(cfg.enc_depth)],
            "dec": [[O.drop_path_scale(draws[2 * cfg.enc_depth + 3 * i + j], pd[i]) for j in range(3)] for i in range(cfg.dec_depth)]}
    order = g["meta/order"].tolist()
    P = tie({k: v.clone().requires_grad_(v.is_floating_point()) for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    o_loss, _ = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"], emulate_bf16=True, drop=drop)
    o_loss.sum().backward()
    o_plain, _ = O.fourm_forward({k: v.detach() for k, v in P.items()}, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"], emulate_bf16=True)
    assert abs(float(o_plain.sum()) - float(o_loss.sum())) > 1e-3                     # the scales matter for this input
    res = {}
    for ckpt in (False, True):
        model.use_act_checkpoint = ckpt
        model.zero_grad(set_to_none=True)
        random.seed(case["order_seed"])
        # (the engine object exists only after the first forward: hand it the uniforms through the class attribute for this call)
        from fourm.hip.engine import FourMEngine
        FourMEngine.drop_uniforms = iter(draws)
        try:
            loss, _ = model(to_device(md), case["N"], case["M"], loss_type=case["loss_type"])
            loss.backward()
        finally:
            FourMEngine.drop_uniforms = None
        torch.cuda.synchronize()
        res[ckpt] = (float(loss.detach()), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None})
    assert abs(res[False][0] - float(o_loss.sum())) < 3e-3 * abs(float(o_loss.sum())), (res[False][0], float(o_loss.sum()))
    errs = []
    for n, p in model.named_parameters():
        og = P[n].grad
        if og is None or float(og.norm()) < 1e-9:
            continue
        errs.append((rel(res[False][1][n], og), n))
    errs.sort(reverse=True)
    assert errs[0][0] < 6e-2 and errs[len(errs) // 2][0] < 2.5e-2, errs[:3]
    record("model.drop_path", case=name, loss_hip=res[False][0], loss_bf16_oracle=float(o_loss.sum()), grad_rel_worst=errs[0][0], grad_rel_median=errs[len(errs) // 2][0])
    assert res[True][0] == res[False][0]
    for n, gr in res[False][1].items():
        assert rel(res[True][1][n], gr) < 2e-5 or float(gr.norm()) < 1e-9, n
    model.eval()
    with torch.no_grad():
        random.seed(case["order_seed"])
        l_eval, _ = model(to_device(md), case["N"], case["M"], loss_type=case["loss_type"])
    assert abs(float(l_eval) - float(o_plain.sum())) < 3e-3 * abs(float(o_plain.sum()))


@pytest.mark.parametrize("name,precision", [("micro_swiglu", "bf16"), ("micro_qknorm", "bf16"), ("micro_swiglu", "fp32")])
def test_zero_attn_model(name, precision):
    """allow_zero_attn=True on every attention module: loss and gradients against the oracle (pinned to upstream in test_model_cpu.py)."""
    import dataclasses
    g, case, model = setup(name)
    if precision == "fp32":
        model.compute_precision, model._engine = "fp32", None
    for mod in model.modules():
        if hasattr(mod, "allow_zero_attn"):
            mod.allow_zero_attn = True
    cfg, md = dataclasses.replace(case["cfg"], zero_attn=True), case["mod_dict"]
    order = g["meta/order"].tolist()
    P = tie({k: v.clone().requires_grad_(v.is_floating_point()) for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    o_loss, _ = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"], emulate_bf16=precision == "bf16")
    o_loss.sum().backward()
    random.seed(case["order_seed"])
    loss, _ = model(to_device(md), case["N"], case["M"], loss_type=case["loss_type"])
    loss.backward()
    tol_l, tol_g = (3e-3, 4.8e-2) if precision == "bf16" else (2e-6, 2e-4)
    assert abs(float(loss.detach()) - float(o_loss.sum())) < tol_l * abs(float(o_loss.sum()))
    worst = max((rel(p.grad, P[n].grad), n) for n, p in model.named_parameters() if P[n].grad is not None and float(P[n].grad.norm()) > 1e-9)
    record("model.zero_attn", case=name, precision=precision, loss_hip=float(loss.detach()), loss_oracle=float(o_loss.sum()), grad_rel_worst=worst[0])
    assert worst[0] < tol_g, worst
