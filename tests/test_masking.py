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
