"""Image-modality masks (SURVEY §8 f3 slice): oracle vs the unmodified upstream UnifiedMasking.image_mask (CPU, build container only),
and the HIP kernel bit-exact against the oracle (GPU)."""
import os
import subprocess
import sys

import pytest
import torch

from oracle import fourm_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CHILD = r'''
import sys, os
sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import ref_stubs; ref_stubs.install()
sys.path.insert(0, "/root/reference")
import torch
from fourm.data.masking import UnifiedMasking
from oracle import fourm_oracle as O
for L_, kin, kt, seed in ((196, 40, 60, 0), (196, 0, 196, 1), (196, 196, 0, 2), (64, 10, None, 3), (196, 128, 68, 4), (16, 3, 0, 5)):
    torch.manual_seed(seed)
    ref = UnifiedMasking.image_mask(None, torch.zeros(L_), L_, kin, kt)
    torch.manual_seed(seed)
    noise = torch.rand(L_)
    im, tm, dam = O.image_mask(noise, kin, kt)
    assert torch.equal(im, ref["input_mask"]) and torch.equal(tm, ref["target_mask"]) and torch.equal(dam, ref["decoder_attention_mask"]), (L_, kin, kt)
    print("ok")
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not on this machine")
def test_image_mask_oracle_matches_upstream():
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert p.stdout.count("ok") == 6


def test_image_mask_oracle_properties():
    g = torch.Generator().manual_seed(0)
    for L_, kin, kt in ((196, 40, 60), (50, 0, 7), (50, 50, 0), (30, 5, None)):
        im, tm, dam = O.image_mask(torch.rand(L_, generator=g), kin, kt)
        n_t = L_ - kin if kt is None else kt
        assert int((~im).sum()) == kin and int((~tm).sum()) == n_t and not bool((~im & ~tm).any())
        assert int(dam.sum()) == n_t and int((dam != 0).sum()) == (1 if n_t else 0)
        if n_t:
            assert int(dam.nonzero()[0]) == int((~tm).nonzero()[0])


@pytest.mark.gpu
@pytest.mark.parametrize("L_", [16, 196, 1024])
@pytest.mark.parametrize("with_target", [True, False])
def test_image_mask_kernel_bit_exact(L_, with_target):
    from fourm.data.masking import image_mask_batched
    g = torch.Generator().manual_seed(L_)
    B = 37
    noise = torch.rand(B, L_, generator=g)
    noise[3, 5] = noise[3, 9]                                    # a tie: lower index first (stable), like the oracle's argsort on CPU
    kin = torch.randint(0, L_ + 1, (B,), generator=g)
    kin[0], kin[1] = 0, L_
    kt = torch.minimum(torch.randint(0, L_ + 1, (B,), generator=g), L_ - kin) if with_target else None
    out = image_mask_batched(L_, kin.cuda(), None if kt is None else kt.cuda(), noise=noise.cuda())
    for b in range(B):
        ids = torch.argsort(noise[b], stable=True)
        im = ids >= int(kin[b])
        tm = ~im if kt is None else ~((ids >= int(kin[b])) & (ids < int(kin[b]) + int(kt[b])))
        o_im, o_tm, o_dam = O.image_mask(noise[b], int(kin[b]), None if kt is None else int(kt[b]))
        if b != 3:                                               # (torch.argsort is not stable by default: compare the tie row to the stable form)
            assert torch.equal(o_im, im) and torch.equal(o_tm, tm)
        dam = torch.zeros(L_, dtype=torch.int32)
        first = int((~tm).nonzero()[0]) if bool((~tm).any()) else 0
        dam[first] = int((~tm).sum())
        assert torch.equal(out["input_mask"][b].cpu(), im) and torch.equal(out["target_mask"][b].cpu(), tm), b
        assert torch.equal(out["decoder_attention_mask"][b].cpu(), dam), b
    out2 = image_mask_batched(L_, kin.cuda(), None if kt is None else kt.cuda(), generator=torch.Generator(device="cuda").manual_seed(1))
    assert int((~out2["input_mask"]).sum()) == int(kin.sum())


def test_pack_mod_dict_host_side():
    """Host half of the compact H2D format: bit order, uint16 ids, the derived decoder_attention_mask is dropped only when it IS derived."""
    from fourm.data import h2d
    g = torch.Generator().manual_seed(0)
    B, L_ = 5, 19
    tm = torch.rand(B, L_, generator=g) < 0.6
    im = torch.rand(B, L_, generator=g) < 0.5
    ids = torch.randint(0, 30000, (B, L_), generator=g)
    dam = h2d._dam_of_target_mask(tm)
    for b in range(B):
        free = (~tm[b]).nonzero().reshape(-1)
        assert int(dam[b].sum()) == len(free) and (len(free) == 0 or int(dam[b, free[0]]) == len(free))
    md = {"tok": {"tensor": ids, "input_mask": im, "target_mask": tm, "decoder_attention_mask": dam},
          "seq": {"tensor": ids.int(), "input_mask": im, "target_mask": tm, "decoder_attention_mask": (~tm).int()}}
    p = h2d.pack_mod_dict(md)
    assert p["tok"]["kind"] == "ids_u16" and p["tok"]["dam"] is None and p["seq"]["dam"] is not None
    bits = p["tok"]["target_mask"].numpy()
    assert bits.shape == (B, 3) and all(((bits[b, i >> 3] >> (i & 7)) & 1) == int(tm[b, i]) for b in range(B) for i in range(L_))
    assert h2d.packed_nbytes(p) < sum(v.numel() * v.element_size() for d in md.values() for v in d.values()) // 2


@pytest.mark.gpu
def test_compact_h2d_round_trip():
    """pack on the host -> unpack on the device reproduces the loader's mod_dict exactly: ids, masks, the image-like
    decoder_attention_mask rebuilt from target_mask, and the normalised pixels bit-identical to to_tensor + normalize."""
    from fourm.data import h2d
    g = torch.Generator().manual_seed(1)
    B = 6
    md, imgs = {}, {}
    for name, L_, V in (("tok_rgb@224", 196, 16384), ("tok_depth@224", 196, 8192)):
        noise = torch.rand(B, L_, generator=g)
        masks = [O.image_mask(noise[b], 30 + b, 50) for b in range(B)]
        md[name] = {"tensor": torch.randint(0, V, (B, 14, 14), generator=g), "input_mask": torch.stack([m[0] for m in masks]),
                    "target_mask": torch.stack([m[1] for m in masks]), "decoder_attention_mask": torch.stack([m[2] for m in masks])}
    md["caption"] = {"tensor": torch.randint(0, 30000, (B, 514), generator=g).int(), "input_mask": torch.rand(B, 514, generator=g) < 0.9,
                     "target_mask": torch.rand(B, 514, generator=g) < 0.9, "decoder_attention_mask": torch.randint(0, 5, (B, 514), generator=g).int()}
    u8 = torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8)
    mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
    rgb = u8.permute(0, 3, 1, 2).float().div(255)                        # to_tensor
    rgb = rgb.sub(mean[None, :, None, None]).div(std[None, :, None, None])   # normalize
    md["rgb@224"] = {"tensor": rgb, "input_mask": torch.rand(B, 196, generator=g) < 0.5, "target_mask": torch.ones(B, 196, dtype=torch.bool),
                     "decoder_attention_mask": torch.zeros(B, 196, dtype=torch.int32)}
    imgs["rgb@224"] = u8
    packed = h2d.pack_mod_dict(md, images_u8=imgs)
    full = sum(v.numel() * v.element_size() for d in md.values() for v in d.values())
    assert h2d.packed_nbytes(packed) < 0.3 * full
    assert packed["tok_rgb@224"]["dam"] is None and packed["caption"]["dam"] is not None
    out = h2d.unpack_mod_dict(packed, "cuda")
    for name, d in md.items():
        for k, v in d.items():
            got = out[name][k].cpu()
            assert got.dtype == v.dtype and got.shape == v.shape, (name, k, got.dtype, v.dtype)
            assert torch.equal(got, v), (name, k)


# ---- token budgets and sequence span masking (second slice of f3) -------------------------------------------------------------------------
import numpy as np  # noqa: E402

from oracle import masking_oracle as MO  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden", "masking.npz")
T_SEQ = 32


def _gold():
    g = np.load(GOLD, allow_pickle=False)
    assert int(g["meta/t_seq"]) == T_SEQ
    return g


def _seq_noise(c, width, checksum):
    """Noise rows of fixture case c, regenerated from the seed the generator used (tests/golden/make_golden_masking.py: seq_noise)."""
    nz = np.random.default_rng(5000 + c).random((T_SEQ, width), dtype=np.float32)
    assert float(nz.astype(np.float64).sum()) == pytest.approx(float(checksum), rel=1e-12), "numpy's random stream differs from the generating environment"
    return nz


def _sentinels(g):
    ids = g["meta/sentinel_ids"]
    return {k: int(v) for k, v in enumerate(ids)}, int(g["meta/pad_id"])


def _seq_case(g, c):
    n = int(g["seq/len"][c])
    ids = g["seq/ids"][c, :n].tolist()
    unit = g["seq/unit"][c, :n].tolist() if bool(g["seq/chunked"][c]) else None
    kt = int(g["seq/tgt_budget"][c])
    return dict(ids=ids, unit=unit, max_tokens=int(g["seq/max_tokens"][c]), kin=int(g["seq/in_budget"][c]), kt=None if kt < 0 else kt, kp=float(g["seq/kp"][c]),
                noise=_seq_noise(c, int(g["seq/noise_width"][c]), g["seq/noise_sum"][c]), r=int(g["seq/r"][c]), voff=int(g["seq/voff"][c]), L=int(g["seq/out_len"][c]))


def test_budget_oracle_matches_upstream_fixture():
    g = _gold()
    for c in range(len(g["bud/n"])):
        b, tries = MO.token_budget(g["bud/main"][c], g["bud/extra"][c], int(g["bud/n"][c]), g["bud/min"][c], g["bud/max"][c])
        assert np.array_equal(b, g["bud/out"][c]) and tries == int(g["bud/tries"][c]), c
        if bool(g["bud/is_target"][c]):
            assert np.array_equal(MO.max_tokens_remaining(g["bud/is_img"], g["bud/max_tokens"], g["bud/min"][c], g["bud/in_budget"][c]), g["bud/max"][c])
    assert {1, 2, 7} <= set(g["bud/tries"].tolist())          # 7 = T + 1: none of the 6 recorded draws met the minimum (upstream keeps the last)


def test_span_oracle_matches_upstream_fixture():
    g = _gold()
    s2i, pad = _sentinels(g)
    kinds = set()
    for c in range(len(g["seq/len"])):
        k = _seq_case(g, c)
        o = MO.sequence_mask(k["ids"], k["max_tokens"], k["kin"], k["kt"], k["kp"], k["noise"], k["r"], s2i, pad, unit_of=k["unit"], vocab_offset=k["voff"])
        assert np.array_equal(o["tensor"], g["seq/tensor"][c, :k["L"]]), c
        assert np.array_equal(o["input_mask"], g["seq/input_mask"][c, :k["L"]]) and np.array_equal(o["target_mask"], g["seq/target_mask"][c, :k["L"]]), c
        assert np.array_equal(o["decoder_attention_mask"], g["seq/dam"][c, :k["L"]]) and o["tries"] == int(g["seq/tries"][c]), c
        kinds.add((k["unit"] is not None, k["kt"] is None, k["kin"] == 0, o["tries"] > 1))
    assert len(kinds) >= 8
    for i in range(int(g["meta/n_emb"])):
        emb = g[f"emb{i}/emb"]
        nz = _seq_noise(100 + i, emb.shape[0], g[f"emb{i}/noise"])
        o = MO.sequence_emb_mask(emb, int(g[f"emb{i}/max_tokens"]), int(g[f"emb{i}/in_budget"]), float(g[f"emb{i}/kp"]), nz, s2i)
        assert np.array_equal(o["tensor"], g[f"emb{i}/tensor"]) and np.array_equal(o["input_mask"], g[f"emb{i}/input_mask"]), i


def test_span_oracle_round_trip():
    """Property (any size): merging the target spans back into the input at their sentinels restores the sequence (what upstream's
    merge_span_masking, text_tokenizer.py:127-137, does at generation time)."""
    rng = np.random.default_rng(3)
    s2i = {k: 1000 + k for k in range(200)}
    sent = set(s2i.values())
    for _ in range(50):
        n = int(rng.integers(1, 200))
        seq = rng.integers(0, 900, n).tolist()
        inp, tgt = MO.span_masking(seq, list(range(n)), rng.random(n, dtype=np.float32), float(rng.random()), s2i)
        spans, cur = {}, None
        for t in tgt:
            if t in sent:
                cur = t; spans[cur] = []
            else:
                spans[cur].append(t)
        merged = [x for t in inp for x in (spans[t] if t in sent else [t])]
        assert merged == seq


@pytest.mark.gpu
def test_token_budget_kernel_bit_exact():
    from fourm.data.masking import token_budgets_batched
    g = _gold()
    dev = "cuda"
    is_t = g["bud/is_target"]
    for target in (False, True):
        sel = np.nonzero(is_t == target)[0]
        main, extra = torch.from_numpy(g["bud/main"][sel]).to(dev), torch.from_numpy(g["bud/extra"][sel]).to(dev)
        # (min_tokens differs between fixture cases: one launch per distinct vector)
        for mn in {tuple(g["bud/min"][c]) for c in sel}:
            rows = [i for i, c in enumerate(sel) if tuple(g["bud/min"][c]) == mn]
            cs = sel[rows]
            out, tries = token_budgets_batched(main[rows], extra[rows], torch.from_numpy(g["bud/n"][cs]).to(dev), list(mn), g["bud/max_tokens"].tolist(),
                                               is_img=g["bud/is_img"] if target else None,
                                               input_budget=torch.from_numpy(g["bud/in_budget"][cs]).to(dev) if target else None)
            assert np.array_equal(out.cpu().numpy(), g["bud/out"][cs]), (target, mn)
            assert np.array_equal(tries.cpu().numpy(), g["bud/tries"][cs])
    # seeded random draws at the benched batch size: kernel == oracle
    gen = torch.Generator().manual_seed(5)
    B, T, M = 256, 5, 21
    alphas = torch.rand(M, generator=gen) * 2 + 0.05
    d = torch.distributions.Dirichlet(alphas)
    torch.manual_seed(11)
    main, extra = d.sample((B, T)), d.sample((B, T, M))
    n = torch.randint(1, 257, (B,), generator=gen)
    mn, mx = torch.randint(0, 4, (M,), generator=gen), torch.randint(8, 257, (M,), generator=gen)
    out, tries = token_budgets_batched(main.to(dev), extra.to(dev), n.to(dev), mn, mx)
    for b in range(B):
        ob, ot = MO.token_budget(main[b].numpy(), extra[b].numpy(), int(n[b]), mn.numpy(), mx.numpy())
        assert np.array_equal(out[b].cpu().numpy(), ob) and int(tries[b]) == ot, b
    assert int(out.sum(1).max()) <= 256 and bool((out.cpu() <= mx[None]).all())


@pytest.mark.gpu
def test_span_mask_kernel_matches_upstream_fixture():
    from fourm.data.masking import sequence_emb_mask_batched, sequence_mask_batched
    g = _gold()
    s2i, pad = _sentinels(g)
    sent = g["meta/sentinel_ids"]
    dev = "cuda"
    for c in range(len(g["seq/len"])):
        k = _seq_case(g, c)
        ids = torch.tensor([k["ids"] + [7] * 3], dtype=torch.int32, device=dev)                 # (row wider than the sequence: len decides)
        unit = None if k["unit"] is None else torch.tensor([k["unit"] + [999] * 3], dtype=torch.int32, device=dev)
        o = sequence_mask_batched(ids, [len(k["ids"])], k["max_tokens"], [k["kin"]], None if k["kt"] is None else [k["kt"]], [k["kp"]],
                                  torch.from_numpy(k["noise"])[None].to(dev), sent, pad, unit=unit, r_choice=[k["r"]], vocab_offset=k["voff"])
        L_ = k["L"]
        assert np.array_equal(o["tensor"][0].cpu().numpy(), g["seq/tensor"][c, :L_]), c
        assert np.array_equal(o["input_mask"][0].cpu().numpy(), g["seq/input_mask"][c, :L_]), c
        assert np.array_equal(o["target_mask"][0].cpu().numpy(), g["seq/target_mask"][c, :L_]), c
        assert np.array_equal(o["decoder_attention_mask"][0].cpu().numpy(), g["seq/dam"][c, :L_]), c
        assert int(o["tries"][0]) == int(g["seq/tries"][c]), c
    for i in range(int(g["meta/n_emb"])):
        emb = g[f"emb{i}/emb"]
        nz = _seq_noise(100 + i, emb.shape[0], g[f"emb{i}/noise"])
        o = sequence_emb_mask_batched(torch.from_numpy(emb)[None].to(dev), int(g[f"emb{i}/max_tokens"]), [int(g[f"emb{i}/in_budget"])], [float(g[f"emb{i}/kp"])],
                                      torch.from_numpy(nz)[None].to(dev), sent)
        assert np.array_equal(o["tensor"][0].cpu().numpy(), g[f"emb{i}/tensor"]) and np.array_equal(o["input_mask"][0].cpu().numpy(), g[f"emb{i}/input_mask"]), i
        assert bool(o["target_mask"].all()) and int(o["decoder_attention_mask"].abs().sum()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("chunked", [False, True])
def test_span_mask_kernel_batch_bit_exact(chunked):
    """A ragged batch at the benched batch size (empty-ish rows, rows longer than max_tokens, zero budgets, no target budget): kernel == oracle."""
    from fourm.data.masking import sequence_mask_batched
    rng = np.random.default_rng(17 + chunked)
    B, W, MT, T = 256, 300, 256, 24
    s2i = {k: 4 + k for k in range(200)}
    sent = np.array([s2i[k] for k in range(200)], dtype=np.int32)
    lens = rng.integers(1, W + 1, B); lens[0], lens[1] = 1, W
    ids = rng.integers(204, 30000, (B, W)).astype(np.int32)
    ids[5, 3] = 10                                                   # a data token that IS a sentinel id: recognised by value, as upstream
    unit = None
    if chunked:
        unit = np.cumsum(rng.random((B, W)) < 0.3, axis=1).astype(np.int32)
        unit -= unit[:, :1]
    kin = rng.integers(0, 129, B); kin[2] = 0
    kt = rng.integers(0, 129, B); kt[3] = -1
    kp = rng.random(B); kp[4] = 1.0; kp[6] = 0.0
    noise = rng.random((B, T, MT), dtype=np.float32)
    r = rng.integers(0, 1 << 30, B).astype(np.int32)
    dev = "cuda"
    o = sequence_mask_batched(torch.from_numpy(ids).to(dev), torch.from_numpy(lens).to(dev), MT, torch.from_numpy(kin).to(dev), torch.from_numpy(kt).to(dev),
                              torch.from_numpy(kp).to(dev), torch.from_numpy(noise).to(dev), sent, 0, unit=None if unit is None else torch.from_numpy(unit).to(dev),
                              r_choice=torch.from_numpy(r).to(dev))
    got = {k: v.cpu().numpy() for k, v in o.items()}
    for b in range(B):
        n = int(lens[b])
        e = MO.sequence_mask(ids[b, :n].tolist(), MT, int(kin[b]), None if kt[b] < 0 else int(kt[b]), float(kp[b]), noise[b], int(r[b]), s2i, 0,
                             unit_of=None if unit is None else unit[b, :n].tolist())
        for k in ("tensor", "input_mask", "target_mask", "decoder_attention_mask"):
            assert np.array_equal(got[k][b], e[k]), (b, k)
        assert int(got["tries"][b]) == e["tries"], b
    # size-independent properties: inputs within budget, targets within budget, the two regions disjoint
    n_in, n_tg = (~got["input_mask"]).sum(1), (~got["target_mask"]).sum(1)
    assert bool((n_in <= kin).all()) and bool((n_tg[kt >= 0] <= kt[kt >= 0]).all()) and not bool((~got["input_mask"] & ~got["target_mask"]).any())


@pytest.mark.gpu
def test_device_unified_masking_contract():
    """DeviceUnifiedMasking over a 4M-style modality_info: shapes / dtypes of the loader contract, budgets respected, and the
    model consumes the result."""
    from fourm.data.masking import DeviceUnifiedMasking
    mk = lambda typ, mx, ia, ta, **kw: dict(type=typ, max_tokens=mx, min_tokens=0, input_alphas=ia, target_alphas=ta, **kw)
    info = {"tok_rgb@224": mk("img", 196, [1.0, 0.5], [1.0, 0.0]), "tok_depth@224": mk("img", 196, [1.0, 0.5], [1.0, 1.0]),
            "caption": mk("seq", 256, [1.0, 5.0], [1.0, 1.0], keep=["random", "all"]), "det": mk("seq", 256, [1.0, 0.05], [1.0, 1.0], keep=["random", "binary"]),
            "tok_global": mk("seq_token", 16, [0.5, 0.0], [0.5, 1.0], vocab_offset=300), "t5_caption": mk("seq_emb", 77, [0.2, 5.0], [0.0, 0.0])}
    s2i = {k: 4 + k for k in range(200)}
    B, dev = 64, "cuda"
    um = DeviceUnifiedMasking(info, None, input_tokens_range=128, target_tokens_range=(64, 128), max_tries=20, device=dev, sentinel_to_id=s2i, pad_id=0)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randint(204, 30000, s, device=dev, generator=g, dtype=torch.int32)
    md = {"tok_rgb": torch.randint(0, 16384, (B, 14, 14), device=dev, generator=g), "tok_depth@224": torch.randint(0, 8192, (B, 14, 14), device=dev, generator=g),
          "caption": {"ids": rnd(B, 80), "len": torch.randint(1, 81, (B,), device=dev, generator=g, dtype=torch.int32)},
          "det": {"ids": rnd(B, 120), "len": torch.full((B,), 120, dtype=torch.int32, device=dev), "unit": (torch.arange(120, device=dev) // 5).int()[None].repeat(B, 1)},
          "tok_global": torch.randint(0, 8192, (B, 16), device=dev, generator=g, dtype=torch.int32), "t5_caption": torch.randn(B, 77, 32, device=dev, generator=g)}
    out = um(md, generator=g)
    assert list(out) == list(info)
    n_in = sum((~out[k]["input_mask"]).flatten(1).sum(1) for k in out)
    n_tg = sum((~out[k]["target_mask"]).flatten(1).sum(1) for k in out)
    assert int(n_in.max()) <= 128 and int(n_tg.max()) <= 128 and int(n_in.sum()) > 0 and int(n_tg.sum()) > 0
    for k, v in out.items():
        Lk = {"img": 196, "seq": 514, "seq_token": 34, "seq_emb": 77}[info[k]["type"]]
        assert v["input_mask"].shape == (B, Lk) and v["input_mask"].dtype == torch.bool and v["decoder_attention_mask"].dtype == torch.int32, k
        assert bool((v["tries"] > 0).all()) if "tries" in v else True
    assert out["t5_caption"]["tensor"].shape == (B, 77, 32) and out["caption"]["tensor"].dtype == torch.int32
    out2 = um(md, generator=torch.Generator(device=dev).manual_seed(0))
    out3 = um(md, generator=torch.Generator(device=dev).manual_seed(0))       # same seed, same draws: the whole pipeline is a function of the generator
    assert all(torch.equal(out2[k][f], out3[k][f]) for k in out2 for f in ("input_mask", "target_mask", "decoder_attention_mask"))


def test_device_masking_check_reports_exhausted_retries_with_the_stock_print(capsys):
    """check(): the 'exhausted retries' warning must print under the stock print() (no ``force`` keyword: that only exists once
    fourm.utils.dist.setup_for_distributed has installed the rank-aware print) and under the rank-aware one on a silent rank."""
    import builtins
    from fourm.data.masking import DeviceUnifiedMasking
    from fourm.utils import dist as fdist
    um = DeviceUnifiedMasking.__new__(DeviceUnifiedMasking)
    assert um.check() == (0, 0)
    um._tries_stat = torch.tensor([0, 3], dtype=torch.int64)
    assert um.check() == (0, 3) and "3 samples exhausted" in capsys.readouterr().out
    assert um._tries_stat.tolist() == [0, 0]
    stock = builtins.print
    try:
        fdist.setup_for_distributed(False)                    # a non-master rank: print is silent unless forced
        um._tries_stat = torch.tensor([0, 2], dtype=torch.int64)
        assert um.check() == (0, 2) and "2 samples exhausted" in capsys.readouterr().out
    finally:
        builtins.print = stock
    um._tries_stat = torch.tensor([1, 0], dtype=torch.int64)
    with pytest.raises(KeyError):
        um.check()


@pytest.mark.gpu
def test_device_masking_feeds_the_model():
    """The loader contract end to end: DeviceUnifiedMasking's batched output is what FourM.forward consumes - loss and
    per-modality losses follow the CPU oracle run on the very same mod_dict."""
    import random
    from fourm.data.masking import DeviceUnifiedMasking
    from oracle import fourm_oracle as FO
    from tests.golden.cases import build_case
    from tests.util_model import build_hip_model, tie
    case = build_case("micro_swiglu")
    cfg = case["cfg"]
    model = build_hip_model(cfg, case["share_embedding"], case["norm_bias"], case["learned_pos"])
    model.load_state_dict(case["sd"], strict=True)
    model = model.cuda().train()
    typ = {"tok": "img", "patch": "img", "seq": "seq", "seq_emb": "seq_emb"}
    info = {m.name: dict(type=typ[m.kind], max_tokens=m.n_pos, min_tokens=0, input_alphas=[1.0, 0.3], target_alphas=[1.0 if m.in_dec else 0.0, 0.5 if m.in_dec else 0.0],
                         keep=["random", "all"]) for m in cfg.mods}
    B, dev = 6, "cuda"
    um = DeviceUnifiedMasking(info, None, input_tokens_range=20, target_tokens_range=18, max_tries=50, device=dev,
                              sentinel_to_id={k: 4 + k for k in range(20)}, pad_id=0)
    g = torch.Generator(device=dev).manual_seed(3)
    raw = {}
    for m in cfg.mods:
        if m.kind == "tok":
            s = int(round(m.n_pos ** 0.5))
            raw[m.name] = torch.randint(0, m.vocab, (B, s, s), device=dev, generator=g)
        elif m.kind == "patch":
            s = int(round(m.n_pos ** 0.5)) * m.patch
            raw[m.name] = torch.randn(B, m.channels, s, s, device=dev, generator=g)
        elif m.kind == "seq":
            raw[m.name] = {"ids": torch.randint(30, m.vocab, (B, m.n_pos), device=dev, generator=g, dtype=torch.int32),
                           "len": torch.randint(2, m.n_pos + 1, (B,), device=dev, generator=g, dtype=torch.int32)}
        else:
            raw[m.name] = torch.randn(B, m.n_pos, m.orig_dim, device=dev, generator=g)
    md = um(raw, generator=g)
    for m in cfg.mods:
        assert md[m.name]["input_mask"].shape == (B, m.tensor_len) and md[m.name]["decoder_attention_mask"].dtype == torch.int32, m.name
    random.seed(5)
    loss, mod_loss = model({k: {a: b for a, b in v.items() if a != "tries"} for k, v in md.items()}, 20, 18)
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is None or bool(torch.isfinite(p.grad).all()) for p in model.parameters())
    cpu_md = {k: {a: b.cpu() for a, b in v.items() if a != "tries"} for k, v in md.items()}
    P = tie({k: v.clone() for k, v in case["sd"].items()}, cfg, case["share_embedding"])
    random.seed(5)
    names = [n for n in cpu_md if n in model.decoder_embeddings]
    order = random.sample(names, len(names))                                  # (cat_decoder_tensors shuffles the same way, fm.py:306)
    with torch.no_grad():
        o_loss, o_mod = FO.fourm_forward(P, cfg, cpu_md, 20, 18, order, emulate_bf16=True)
    assert abs(float(loss.detach()) - float(o_loss.sum())) < 5e-3 * abs(float(o_loss.sum())), (float(loss.detach()), float(o_loss.sum()))


def test_device_masking_for_a_named_model():
    """Host logic (no GPU): the DeviceUnifiedMasking built for a model mirrors the registry - modality types, max_tokens, one Dirichlet
    component over the encoder / decoder sides, the tokenizer's sentinel ids - and the output row lengths are the loader contract's."""
    from fourm.data.modality_info import MODALITY_INFO
    from fourm.data.synthetic import device_masking_for, modality_shapes
    from tests.golden.cases import build_case
    from tests.util_model import build_hip_model
    case = build_case("ti_mod7")
    model = build_hip_model(case["cfg"], case["share_embedding"], case["norm_bias"], case["learned_pos"])
    um = device_masking_for(model, 128, 128, device="cpu")
    shp = modality_shapes(model)
    assert list(um.modality_info) == list(shp)
    for n, info in um.modality_info.items():
        ref = MODALITY_INFO[n]
        assert info["type"] == ref["type"] and info["max_tokens"] == (ref["max_tokens"] or 196), n        # (registry: None = (224 / 16)^2 grid tokens)
        assert (info["input_alphas"][0] > 0) == (n in model.encoder_embeddings) and (info["target_alphas"][0] > 0) == (n in model.decoder_embeddings)
        L_out = {"img": info["max_tokens"], "seq": 2 * (info["max_tokens"] + 1), "seq_emb": info["max_tokens"]}[info["type"]]
        assert L_out == shp[n]["L"], n
    assert um.sentinel_ids.tolist() == list(range(4, 204)) and um.pad_id == 0 and um.num_dirichlets == 1
    assert float(um.target_alphas[0][list(shp).index("rgb@224")]) < 1e-6          # rgb pixels are never a target (clamped alpha, masking.py:160)
    with pytest.raises(ValueError):
        from fourm.data import SyntheticLoader
        SyntheticLoader(model, 2, 128, 128, 1, device="cpu", masking="dirichlet")       # the producer needs the GPU: no CPU fallback
