"""Host-side checks that need no GPU: parameter tree / state_dict layout against the upstream fixtures,
registry, config parsing, optimizer grouping, checkpoint layout, C-ABI export list."""
import ctypes
import os
import re
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests.golden.cases import CASES, build_case
from tests.util_model import build_hip_model

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", list(CASES))
def test_state_dict_layout_matches_upstream(name):
    """Keys, shapes, parameter-vs-buffer split and tying are those of the upstream model (fixture meta)."""
    g = np.load(os.path.join(GOLD, f"{name}.npz"))
    case = build_case(name)
    model = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    sd = model.state_dict()
    assert list(sd.keys()) == g["meta/keys"].tolist()
    assert [",".join(map(str, v.shape)) for v in sd.values()] == g["meta/shapes"].tolist()
    assert [k for k, _ in model.named_parameters()] == g["meta/param_keys"].tolist()
    assert [k for k, _ in model.named_buffers()] == g["meta/buffer_keys"].tolist()
    missing = model.load_state_dict(case["sd"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    # fixed position tables are regenerated identically to upstream's
    for k, v in model.state_dict().items():
        if k.endswith("pos_emb"):
            assert torch.equal(v, case["sd"][k]), k


def test_named_factories_and_registry():
    from fourm.utils import create_model, list_models
    from fourm.data.modality_info import MODALITY_INFO
    names = list_models("fm_*")
    assert len(names) == 13 and "fm_base_12e_12d_swiglu_nobias" in names
    mods_in = ["rgb@224", "tok_depth@224", "caption"]
    enc = {m: MODALITY_INFO[m]["encoder_embedding"](**(dict(patch_size=16, image_size=224) if MODALITY_INFO[m]["type"] == "img" else {}))
           for m in mods_in}
    dec = {m: MODALITY_INFO[m]["decoder_embedding"](**(dict(patch_size=16, image_size=224) if MODALITY_INFO[m]["type"] == "img" else {}))
           for m in ["tok_depth@224", "caption"]}
    model = create_model("fm_tiny_6e_6d_swiglu_nobias", encoder_embeddings=enc, decoder_embeddings=dec,
                         modality_info={m: MODALITY_INFO[m] for m in mods_in}, num_register_tokens=None)
    assert model.dim == 384 and len(model.encoder) == 6 and model.encoder[0].mlp.fc1.weight.shape == (1024, 384)
    assert model.decoder_embeddings["caption"].to_logits.weight is model.decoder_embeddings["caption"].token_emb.weight
    assert model.decoder_embeddings["caption"].mod_emb is model.encoder_embeddings["caption"].mod_emb
    assert model.encoder_embeddings["caption"].pos_emb.shape == (1, 512, 384)
    assert MODALITY_INFO["caption"]["id"] == 32652 and MODALITY_INFO["tok_rgb@224"]["id"] == 11606


def test_fm_config_wrapper():
    from fourm.models.fm import FM
    cfg = dict(domains_in=["rgb@224", "caption"], domains_out=["caption", "tok_rgb@224"], image_size=224, patch_size=16,
               norm_bias=False, act_layer="SiLU", dim=384, encoder_depth=1, decoder_depth=1, num_heads=6, mlp_ratio=4,
               qkv_bias=False, proj_bias=False, mlp_bias=False, gated_mlp=True)
    m = FM(cfg)
    assert "encoder.0.norm1.bias" in dict(m.named_buffers())
    assert m.decoder_embeddings["caption"].to_logits.weight is not m.decoder_embeddings["caption"].token_emb.weight


def test_cpu_forward_fails_loudly():
    case = build_case("micro_swiglu")
    model = build_hip_model(case["cfg"])
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        model(case["mod_dict"], case["N"], case["M"])


def test_optimizer_groups_and_checkpoint_layout(tmp_path):
    from fourm.utils import create_optimizer, save_model, auto_load_model, NativeScalerWithGradNormCount
    case = build_case("micro_gelu")
    model = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    args = SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95], momentum=0.9,
                           output_dir=str(tmp_path), auto_resume=True, resume="", start_epoch=0)
    opt = create_optimizer(args, model)
    assert isinstance(opt, torch.optim.AdamW)
    groups = {g["weight_decay"]: g for g in opt.param_groups}
    assert set(groups) == {0.0, 0.05} and all("lr_scale" in g for g in opt.param_groups)
    nd = {id(p) for p in groups[0.0]["params"]}
    for n, p in model.named_parameters():
        assert (id(p) in nd) == ("norm." in n or ".norm" in n or n.endswith(".bias")), n
    scaler = NativeScalerWithGradNormCount(enabled=False)
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    opt.step()
    save_model(args, 3, model, model, opt, scaler)
    blob = torch.load(tmp_path / "checkpoint-3.pth", weights_only=False)
    assert set(blob) == {"model", "epoch", "args", "scaler", "optimizer"} and blob["epoch"] == 3
    st = blob["optimizer"]["state"][0]
    assert set(st) == {"step", "exp_avg", "exp_avg_sq"}
    model2 = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    opt2 = create_optimizer(args, model2)
    auto_load_model(args, model2, model2, opt2, scaler)
    assert args.start_epoch == 4
    for a, b in zip(model.state_dict().values(), model2.state_dict().values()):
        assert torch.equal(a, b)
    # model_ema is stored and restored like upstream does (checkpoint.py:113-114,154-156), never dropped silently
    import copy
    ema = SimpleNamespace(ema=copy.deepcopy(model))
    with torch.no_grad():
        for p in ema.ema.parameters():
            p.mul_(0.5)
    save_model(args, 5, model, model, opt, scaler, model_ema=ema)
    assert "model_ema" in torch.load(tmp_path / "checkpoint-5.pth", weights_only=False)
    args.resume, args.model_ema = str(tmp_path / "checkpoint-5.pth"), True
    ema2 = SimpleNamespace(ema=copy.deepcopy(model2))
    auto_load_model(args, model2, model2, opt2, scaler, model_ema=ema2)
    for a, b in zip(ema.ema.state_dict().values(), ema2.ema.state_dict().values()):
        assert torch.equal(a, b)
    with pytest.raises(ValueError, match="model_ema"):
        auto_load_model(args, model2, model2, opt2, scaler, model_ema=None)


def test_c_abi_exports_every_declared_symbol():
    """libfourm_hip.so loads without a GPU and exports exactly what include/fourm_hip.h declares."""
    from fourm.hip import _lib
    header = open(os.path.join(ROOT, "include", "fourm_hip.h")).read()
    declared = set(re.findall(r"\b(fm_[a-z0-9_]+)\s*\(", header))
    declared -= {"fm_last_error"} - {"fm_last_error"}
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(_lib.lib, sym), sym
    assert _lib.lib.fm_abi_version() == int(re.search(r"#define FM_ABI_VERSION (\d+)", header).group(1))
    # enum values used by the Python side
    for name, val in dict(FM_EPI_SWIGLU=_lib.EPI_SWIGLU, FM_MASK_DECODER=_lib.MASK_DECODER, FM_KIND_SEQ_EMB=_lib.KIND_SEQ_EMB,
                          FM_LOSS_TOKEN=_lib.LOSS_TOKEN, FM_MAX_MODS=_lib.FM_MAX_MODS).items():
        assert re.search(rf"{name}\s*(=|\s)\s*{val}\b", header), name
    # struct sizes agree with the C compiler's layout rules (natural alignment, no packing)
    assert ctypes.sizeof(_lib.GemmGroup) == 32 and ctypes.sizeof(_lib.ModDesc) % 8 == 0


def test_gemm_nt4_keeps_the_agpr_file_to_itself(tmp_path):
    """gemm_nt4.hip names 256 accumulator registers (the AGPR file) in asm text the compiler cannot see into.  Sound only while the compiler
    itself never uses an AGPR in those kernels (it would, as spill space, once the VGPR file fills): tools/check_nt4_asm.py compiles the file to
    assembly and checks every gemm_nt4 kernel - no AGPR reference outside the asm statements, nothing in scratch."""
    import shutil
    import sys
    if not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None:
        pytest.skip("no hipcc")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_nt4_asm
    kernels, problems = check_nt4_asm.check(check_nt4_asm.compile_to_asm(str(tmp_path)))
    assert not problems, problems[:5]
    assert len(kernels) >= 2 and all(k["mfma"] > 100 and k["asm_agpr"] > 100 for k in kernels.values()), kernels
    # gemm_tn4.hip (the TN job list on 256 x 384 tiles): the same contract
    kernels, problems = check_nt4_asm.check(check_nt4_asm.compile_to_asm(str(tmp_path), "gemm_tn4.hip"), "gemm_tn4_multi_kernel")
    assert not problems, problems[:5]
    assert len(kernels) == 1 and all(k["mfma"] >= 96 and k["asm_agpr"] > 100 for k in kernels.values()), kernels


def test_ctypes_mirrors_match_the_header_layout(tmp_path):
    """Every struct of include/fourm_hip.h: sizeof and the offset of every field, as gcc lays them out, against the
    ctypes mirrors the Python side passes to libfourm_hip.so (a silent mismatch would corrupt arguments)."""
    import shutil
    import subprocess
    from fourm.hip import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {"fm_gemm_group": _lib.GemmGroup, "fm_gemm_nt_args": _lib.GemmNTArgs, "fm_gemm_tn_args": _lib.GemmTNArgs, "fm_gemm_tn_job": _lib.GemmTNJob,
             "fm_gemm_f32_args": _lib.GemmF32Args, "fm_adamw_job": _lib.AdamWJob,
             "fm_attn_args": _lib.AttnArgs, "fm_mod_desc": _lib.ModDesc, "fm_select_desc": _lib.SelectDesc,
             "fm_embed_bwd_mod": _lib.EmbedBwdMod, "fm_embed_bwd_desc": _lib.EmbedBwdDesc, "fm_shadow_desc": _lib.ShadowDesc, "fm_fold_grad_job": _lib.FoldGradJob,
             "fm_span_mask_args": _lib.SpanMaskArgs}
    header = open(os.path.join(ROOT, "include", "fourm_hip.h")).read()
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "fourm_hip.h"', 'int main(void) {']
    fields = {}
    for cname in pairs:
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), header, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                nm = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part.strip()).group(1)
                names.append(nm)
        fields[cname] = names
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for nm in names:
            lines.append(f'  printf("{cname}.{nm} %zu\\n", offsetof({cname}, {nm}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert ctypes.sizeof(cls) == int(out[cname]), (cname, ctypes.sizeof(cls), out[cname])
        mirror = [f[0] for f in cls._fields_]
        assert mirror == fields[cname], (cname, mirror, fields[cname])
        for nm in mirror:
            assert getattr(cls, nm).offset == int(out[f"{cname}.{nm}"]), (cname, nm)


def test_ctypes_prototypes_match_the_header():
    """Number and kind (pointer / int / float) of parameters of every fm_* entry point: header vs the ctypes argtypes."""
    from fourm.hip import _lib
    header = open(os.path.join(ROOT, "include", "fourm_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    protos = re.findall(r"\b(?:int|void|const char\*)\s+(fm_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", header)
    assert len(protos) >= 30
    for name, params in protos:
        fn = getattr(_lib.lib, name)
        plist = [p.strip() for p in params.split(",") if p.strip() and p.strip() != "void"]
        if fn.argtypes is None:
            assert name in ("fm_abi_version", "fm_last_error", "fm_get_gemm_nt_config", "fm_get_gemm_tn_config", "fm_get_tn_transpose_read",
                            "fm_get_attn_transpose_read", "fm_set_gemm_nt_config", "fm_set_attn_transpose_read"), name
            continue
        assert len(fn.argtypes) == len(plist), (name, len(fn.argtypes), plist)
        for at, decl in zip(fn.argtypes, plist):
            if "*" in decl:
                assert at in (ctypes.c_void_p,) or hasattr(at, "_type_"), (name, decl, at)
            elif decl.startswith("float"):
                assert at is ctypes.c_float, (name, decl, at)
            elif decl.startswith("int64_t"):
                assert at is ctypes.c_int64, (name, decl, at)
            else:
                assert at in (ctypes.c_int, ctypes.c_int32), (name, decl, at)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not on this machine")
def test_oracle_drop_path_matches_upstream():
    """Stochastic depth: the oracle's explicit per-branch scales against the unmodified upstream blocks whose DropPath draws are replayed
    (fm_utils.py:64-87, :331-334, :362-366)."""
    import subprocess
    import sys
    child = r'''
import sys, os, random
sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import make_golden as MG
import torch
from oracle import fourm_oracle as O
from tests.golden.cases import build_case
case = build_case("micro_swiglu")
cfg, sd, md = case["cfg"], case["sd"], case["mod_dict"]
model = MG.upstream_model(cfg, case["share_embedding"], case["norm_bias"], case["learned_pos"])
model.load_state_dict(sd, strict=True)
model.train()
B = next(iter(md.values()))["tensor"].shape[0]
pe = [0.2 + 0.1 * i for i in range(cfg.enc_depth)]
pd = [0.3 + 0.1 * i for i in range(cfg.dec_depth)]
for blk, p in zip(list(model.encoder) + list(model.decoder), pe + pd):
    blk.drop_path = MG.ref_utils.DropPath(p)
g = torch.Generator().manual_seed(4)
draws = [torch.rand(B, generator=g) for _ in range(2 * cfg.enc_depth + 3 * cfg.dec_depth)]
it = iter(draws)
real = torch.rand
def fake(*a, **k):
    shape = a[0] if a and isinstance(a[0], (tuple, list, torch.Size)) else a
    if len(shape) == 3 and tuple(shape[1:]) == (1, 1):
        return next(it).reshape(shape).to(k.get("dtype", torch.float32))
    return real(*a, **k)
torch.rand = fake
random.seed(case["order_seed"])
loss, _ = model(MG.clone_mod_dict(md), case["N"], case["M"], loss_type=case["loss_type"])
loss.sum().backward()
torch.rand = real
assert next(it, None) is None
drop = {"enc": [[O.drop_path_scale(draws[2 * i + j], pe[i]) for j in range(2)] for i in range(cfg.enc_depth)],
        "dec": [[O.drop_path_scale(draws[2 * cfg.enc_depth + 3 * i + j], pd[i]) for j in range(3)] for i in range(cfg.dec_depth)]}
assert any(float(s.min()) == 0.0 for side in drop.values() for l in side for s in l) and any(float(s.max()) > 1.0 for side in drop.values() for l in side for s in l)
P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
for m in cfg.mods:
    if m.in_enc and m.in_dec:
        P[f"decoder_embeddings.{m.name}.mod_emb"] = P[f"encoder_embeddings.{m.name}.mod_emb"]
    if m.in_dec and case["share_embedding"]:
        P[f"decoder_embeddings.{m.name}.to_logits.weight"] = P[f"decoder_embeddings.{m.name}.token_emb.weight"]
order = MG.dec_order_for_seed([n for n in md if n in model.decoder_embeddings], case["order_seed"])
ol, _ = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"], drop=drop)
ol.sum().backward()
assert abs(float(ol.sum()) - float(loss.sum())) < 2e-5 * abs(float(loss.sum())), (float(ol.sum()), float(loss.sum()))
n = 0
for k, p in model.named_parameters():
    if p.grad is not None and P[k].grad is not None and float(p.grad.norm()) > 1e-8:
        assert float((P[k].grad - p.grad).norm() / p.grad.norm()) < 1e-3, k
        n += 1
assert n > 20
print("ok")
'''
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + child], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and "ok" in p.stdout, (p.stdout + p.stderr)[-3000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not on this machine")
def test_oracle_zero_attn_matches_upstream():
    """allow_zero_attn (softmax1, fm_utils.py:28-30): oracle vs the unmodified upstream attention modules with the flag set."""
    import subprocess
    import sys
    child = r'''
import sys, os, random
sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import make_golden as MG
import dataclasses, torch
from oracle import fourm_oracle as O
from tests.golden.cases import build_case
for name in ("micro_swiglu", "micro_qknorm"):
    case = build_case(name)
    cfg, sd, md = dataclasses.replace(case["cfg"], zero_attn=True), case["sd"], case["mod_dict"]
    model = MG.upstream_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(sd, strict=True)
    n = 0
    for mod in model.modules():
        if hasattr(mod, "allow_zero_attn"):
            mod.allow_zero_attn = True; n += 1
    assert n == len(model.encoder) + 2 * len(model.decoder)
    model.train()
    random.seed(case["order_seed"])
    loss, _ = model(MG.clone_mod_dict(md), case["N"], case["M"], loss_type=case["loss_type"])
    loss.sum().backward()
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    for m in cfg.mods:
        if m.in_enc and m.in_dec:
            P[f"decoder_embeddings.{m.name}.mod_emb"] = P[f"encoder_embeddings.{m.name}.mod_emb"]
        if m.in_dec and case["share_embedding"]:
            P[f"decoder_embeddings.{m.name}.to_logits.weight"] = P[f"decoder_embeddings.{m.name}.token_emb.weight"]
    order = MG.dec_order_for_seed([n_ for n_ in md if n_ in model.decoder_embeddings], case["order_seed"])
    ol, _ = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"])
    ol.sum().backward()
    plain, _ = O.fourm_forward({k: v.detach() for k, v in P.items()}, case["cfg"], md, case["N"], case["M"], order, loss_type=case["loss_type"])
    assert abs(float(plain.sum()) - float(ol.sum())) > 1e-6                           # the flag changes the result
    assert abs(float(ol.sum()) - float(loss.sum())) < 2e-5 * abs(float(loss.sum())), (float(ol.sum()), float(loss.sum()))
    for k, p in model.named_parameters():
        if p.grad is not None and P[k].grad is not None and float(p.grad.norm()) > 1e-8:
            assert float((P[k].grad - p.grad).norm() / p.grad.norm()) < 1e-3, k
    print("ok")
'''
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + child], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0 and p.stdout.count("ok") == 2, (p.stdout + p.stderr)[-3000:]


def test_context_norm_hoist_algebra():
    """The derivation behind the engine's context-norm hoist (fourm/hip/engine.py hoist_ctx), in plain fp64 torch: with x_hat = (c - mean) rstd,
        kv_l = LN_l(c) W_l^T = x_hat (W_l diag(gamma_l))^T                                    (bias-free norms),
        dL/dW_l = dL/dW'_l diag(gamma_l),   dL/dgamma_l = column sums of dL/dW'_l * W_l          (W'_l = W_l diag(gamma_l)),
        dL/dc = LayerNorm backward (no affine part) of  sum_l dkv_l W'_l  = [dkv_0 | dkv_1 | ...] [W'_0; W'_1; ...]
    against autograd through L separate LayerNorms of the same context (what upstream's decoder blocks compute, fm_utils.py:364)."""
    torch.manual_seed(0)
    R, D, L = 24, 16, 3
    c = torch.randn(R, D, dtype=torch.float64, requires_grad=True)
    Ws = [torch.randn(2 * D, D, dtype=torch.float64, requires_grad=True) for _ in range(L)]
    gs = [(torch.rand(D, dtype=torch.float64) + 0.5).requires_grad_(True) for _ in range(L)]
    up = [torch.randn(R, 2 * D, dtype=torch.float64) for _ in range(L)]                          # d(loss) / d(kv_l)
    eps = 1e-6
    loss = sum((torch.nn.functional.layer_norm(c, (D,), g, None, eps) @ W.t() * u).sum() for W, g, u in zip(Ws, gs, up))
    loss.backward()
    with torch.no_grad():
        mu, var = c.mean(-1, keepdim=True), c.var(-1, unbiased=False, keepdim=True)
        rstd = (var + eps).rsqrt()
        xh = (c - mu) * rstd
        dxh = torch.cat(up, 1) @ torch.cat([W * g[None, :] for W, g in zip(Ws, gs)], 0)          # one GEMM over all layers
        dc = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))   # LayerNorm backward, unit weight
        assert torch.allclose(dc, c.grad, rtol=1e-9, atol=1e-11)
        for W, g, u in zip(Ws, gs, up):
            dWp = u.t() @ xh                                                                     # the ordinary weight-gradient GEMM, on x_hat
            assert torch.allclose(dWp * g[None, :], W.grad, rtol=1e-9, atol=1e-11)
            assert torch.allclose((dWp * W).sum(0), g.grad, rtol=1e-9, atol=1e-11)


def test_every_environment_switch_is_documented():
    """INTEGRATION.md section D lists every FOURM_* variable the package or bench.py reads (a switch that exists only in the source is a
    behaviour nobody can look up)."""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    files = [f for f in glob.glob(os.path.join(root, "ml-4m_amd", "**", "*"), recursive=True) if f.endswith((".py", ".hip", ".cpp", ".h"))]
    for f in files + [os.path.join(root, "bench.py")]:
        with open(f, errors="ignore") as fh:
            names |= set(re.findall(r'(?:getenv\(|environ\.get\(|environ\[)\s*"(FOURM_[A-Z0-9_]+)"', fh.read()))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    missing = sorted(n for n in names if n not in doc)
    assert len(names) > 20 and not missing, missing
