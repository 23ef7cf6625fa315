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
