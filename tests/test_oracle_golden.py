"""The CPU oracle replayed against the fixtures produced by the unmodified upstream model
(tests/golden/make_golden.py).  Runs anywhere: no reference tree, no GPU."""
import os

import numpy as np
import pytest
import torch

from oracle import fourm_oracle as O
from tests.golden.cases import CASES, build_case

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, f"{name}.npz"), allow_pickle=False)


def tie(P, cfg, share):
    for m in cfg.mods:
        if m.in_enc and m.in_dec:
            P[f"decoder_embeddings.{m.name}.mod_emb"] = P[f"encoder_embeddings.{m.name}.mod_emb"]
        if m.in_dec and share:
            P[f"decoder_embeddings.{m.name}.to_logits.weight"] = P[f"decoder_embeddings.{m.name}.token_emb.weight"]
    return P


# (cases marked `big` — batch 256 — take minutes and tens of GB through the CPU oracle; their fixtures are compared with the HIP
#  path in tests/test_b256_golden_gpu.py and with the oracle's integer selection in test_big_fixture_selection_is_exact below)
@pytest.mark.parametrize("name", [n for n in CASES if not CASES[n].get("big")])
def test_oracle_matches_upstream_fixture(name):
    g = load(name)
    case = build_case(name)
    cfg, sd, md = case["cfg"], case["sd"], case["mod_dict"]
    # guards against RNG drift between the generating and the replaying environment
    wsum = sum(float(v.double().abs().sum()) for v in sd.values())
    isum = sum(float(t.double().abs().sum()) for d in md.values() for t in d.values())
    assert wsum == pytest.approx(float(g["meta/weight_checksum"]), rel=1e-9)
    assert isum == pytest.approx(float(g["meta/input_checksum"]), rel=1e-9)
    # state_dict layout (keys, order-insensitive; shapes)
    shapes = dict(zip(g["meta/keys"].tolist(), g["meta/shapes"].tolist()))
    assert set(shapes) == set(sd)
    for k, v in sd.items():
        assert ",".join(map(str, v.shape)) == shapes[k], k

    P = tie({k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}, cfg, case["share_embedding"])
    order = g["meta/order"].tolist()
    taps = {}
    loss, mod_loss = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, loss_type=case["loss_type"], taps=taps)
    # integer / boolean outputs: bit-exact
    assert np.array_equal(taps["enc_mask"].numpy(), g["enc/mask"])
    assert np.array_equal(taps["enc_mod_mask"].numpy(), g["enc/mod_mask"])
    assert np.array_equal(taps["dec_mask"].numpy(), g["dec/mask"])
    assert np.array_equal(taps["dec_mod_mask"].numpy(), g["dec/mod_mask"])
    assert np.array_equal(taps["dec_target_ids"].numpy(), g["dec/target_ids"])
    assert np.array_equal(np.packbits(taps["dec_attn_mask"].numpy(), axis=-1), g["dec/attn_mask"])
    # gathered rows are copies: exact
    if "enc/tokens" in g:
        assert np.array_equal(taps["enc_tokens"].detach().numpy(), g["enc/tokens"])
        assert np.array_equal(taps["enc_emb"].detach().numpy(), g["enc/emb"])
        assert np.array_equal(taps["dec_tokens"].detach().numpy(), g["dec/tokens"])
        assert np.array_equal(taps["dec_emb"].detach().numpy(), g["dec/emb"])
    else:
        np.testing.assert_allclose(taps["enc_tokens"].detach().sum(-1).numpy(), g["enc/tokens_rowsum"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(taps["dec_emb"].detach().sum(-1).numpy(), g["dec/emb_rowsum"], rtol=1e-5, atol=1e-5)
    # floating point: fp32 vs fp32, tolerance 2e-5 relative
    np.testing.assert_allclose(loss.detach().reshape(-1).numpy(), g["loss"], rtol=2e-5)
    for k, v in mod_loss.items():
        np.testing.assert_allclose(v.detach().reshape(-1).numpy(), g[f"mod_loss/{k}"], rtol=2e-5, atol=1e-6)
    loss.sum().backward()
    for key in g.files:
        if key.startswith("grad_l2/"):
            k = key[len("grad_l2/"):]
            assert P[k].grad is not None, k
            assert float(P[k].grad.double().norm()) == pytest.approx(float(g[key]), rel=5e-4, abs=1e-7), k
            np.testing.assert_allclose(P[k].grad.reshape(-1)[:16].numpy(), g[f"grad_head/{k}"], rtol=2e-3, atol=1e-6)
    with torch.no_grad():
        logits = O.fourm_forward(P, cfg, md, case["N"], case["M"], order, return_logits=True)
    for k, v in logits.items():
        assert float(v.double().norm()) == pytest.approx(float(g[f"logits_fro/{k}"]), rel=1e-5)


@pytest.mark.parametrize("name", [n for n in CASES if CASES[n].get("big")])
def test_big_fixture_selection_is_exact(name):
    """Batch-256 fixture: checksums, and the oracle's token selection alone (fm.py:343-449; integers and gathered rows) — the
    trunk at this size is left to the GPU test."""
    g = load(name)
    case = build_case(name)
    cfg, sd, md = case["cfg"], case["sd"], case["mod_dict"]
    assert sum(float(v.double().abs().sum()) for v in sd.values()) == pytest.approx(float(g["meta/weight_checksum"]), rel=1e-9)
    assert sum(float(t.double().abs().sum()) for d in md.values() for t in d.values()) == pytest.approx(float(g["meta/input_checksum"]), rel=1e-9)
    P = tie(dict(sd), cfg, case["share_embedding"])
    with torch.no_grad():
        enc = O.select_encoder(P, cfg, md, case["N"], O._Num(False))
        dec = O.select_decoder(P, cfg, md, case["M"], g["meta/order"].tolist())
    assert np.array_equal(enc["mask"].numpy(), g["enc/mask"])
    assert np.array_equal(enc["mod_mask"].numpy(), g["enc/mod_mask"])
    assert np.array_equal(dec["mask"].numpy(), g["dec/mask"])
    assert np.array_equal(dec["mod_mask"].numpy(), g["dec/mod_mask"])
    assert np.array_equal(dec["target_ids"].numpy(), g["dec/target_ids"])
    assert np.array_equal(np.packbits(dec["attn_mask"].numpy(), axis=-1), g["dec/attn_mask"])
    np.testing.assert_allclose(enc["tokens"].sum(-1).numpy(), g["enc/tokens_rowsum"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dec["emb"].sum(-1).numpy(), g["dec/emb_rowsum"], rtol=1e-5, atol=1e-5)


def test_partition_equals_float_argsort():
    """The prefix-sum partition is what fm.py:364-367's float argsort computes, also at the largest
    concatenated lengths the named configs reach (2204 / 4481)."""
    g = torch.Generator().manual_seed(0)
    for L in (7, 2204, 4481):
        mask = torch.rand(5, L, generator=g) < 0.9
        ref = torch.argsort(mask + torch.arange(L)[None] * 1e-6, dim=1)
        for n in (1, min(L, 128), L):
            assert torch.equal(O.stable_partition_keep(mask, n), ref[:, :n])


def test_emulated_bf16_close_to_fp32():
    case = build_case("micro_swiglu")
    g = load("micro_swiglu")
    P = tie({k: v.clone() for k, v in case["sd"].items()}, case["cfg"], True)
    order = g["meta/order"].tolist()
    with torch.no_grad():
        a, _ = O.fourm_forward(P, case["cfg"], case["mod_dict"], case["N"], case["M"], order)
        b, _ = O.fourm_forward(P, case["cfg"], case["mod_dict"], case["N"], case["M"], order, emulate_bf16=True)
    assert abs(float(a) - float(b)) < 5e-2 and float(a) != float(b)
