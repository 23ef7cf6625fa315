"""BASELINE.json configs[1] at full size (4M-B mod7, per-GPU batch 256, 128 + 128 tokens) through properties that do
not need the CPU oracle (it takes minutes at this size):
  * selection: kept slots are the first valid positions in order (a stable partition), masks / ids consistent;
  * the forward is deterministic (bit-identical loss and per-modality losses across runs);
  * loss at random init ~ the mean log-vocabulary (every head starts near uniform);
  * backward is linear in the upstream gradient, and gradients accumulate additively;
  * one AdamW step with weight decay 0 moves every parameter by at most lr (|m_hat / sqrt(v_hat)| <= 1 at step 1)."""
import math
import os
import random
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def job():
    import bench
    from fourm.data.synthetic import synthetic_batch
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    model = bench.build_model("fm_base_12e_12d_swiglu_nobias", dev).train()
    batch = synthetic_batch(model, 256, 128, 128, device=dev, seed=7)
    return model, batch


def flat_grads(model):
    model.engine._ensure_flat()
    return model.engine.flat_grads.detach().clone()


def test_selection_is_a_stable_partition(job):
    model, batch = job
    with torch.no_grad():
        tok, emb, mask, mod = model.forward_mask_encoder(batch, 128)
        cat_tok, cat_emb, cat_mask, cat_mod = model.cat_encoder_tensors(batch)
    B = tok.shape[0]
    assert tok.shape == (B, 128, 768) and mask.shape == (B, 1, 128)
    valid = ~cat_mask                                              # (B, O)
    n_valid = valid.sum(1)
    kept_valid = (~mask[:, 0]).sum(1)
    assert torch.equal(kept_valid, torch.clamp(n_valid, max=128))   # every valid position is kept, up to the budget
    order = torch.argsort((~valid).int(), dim=1, stable=True)[:, :128]      # valid positions first, original order
    want_tok = torch.gather(cat_tok, 1, order[..., None].expand(-1, -1, 768)).masked_fill(mask[:, 0, :, None], 0.0)
    want_emb = torch.gather(cat_emb, 1, order[..., None].expand(-1, -1, 768)).masked_fill(mask[:, 0, :, None], 0.0)
    assert torch.equal(tok, want_tok) and torch.equal(emb, want_emb)
    want_mod = torch.gather(cat_mod, 1, order).masked_fill(mask[:, 0], -1)
    assert torch.equal(mod, want_mod)


def test_forward_is_deterministic_and_starts_near_uniform(job):
    model, batch = job
    with torch.no_grad():
        random.seed(3); l1, m1 = model(batch, 128, 128)
        random.seed(3); l2, m2 = model(batch, 128, 128)
    assert torch.equal(l1, l2) and all(torch.equal(m1[k], m2[k]) for k in m1)
    for name, v in m1.items():
        vocab = model.decoder_embeddings[name].vocab_size
        assert abs(float(v) - math.log(vocab)) < 0.35, (name, float(v), math.log(vocab))
    assert abs(float(l1) - sum(float(v) for v in m1.values()) / len(m1)) < 1e-5


def test_backward_is_linear_and_accumulates(job):
    model, batch = job
    model.zero_grad(set_to_none=True)
    random.seed(3); loss, _ = model(batch, 128, 128); loss.backward()
    g1 = flat_grads(model)
    assert torch.isfinite(g1).all() and float(g1.abs().sum()) > 0
    model.zero_grad(set_to_none=True)
    random.seed(3); loss, _ = model(batch, 128, 128); (0.5 * loss).backward()
    gh = flat_grads(model)
    err = float((gh - 0.5 * g1).norm() / (0.5 * g1).norm())
    assert err < 2e-3, err                                          # bf16 rounding of the scaled d(logits) + atomic order
    random.seed(3); loss, _ = model(batch, 128, 128); (0.5 * loss).backward()      # accumulates onto gh
    err = float((flat_grads(model) - g1).norm() / g1.norm())
    assert err < 2e-3, err


def test_adamw_first_step_is_bounded_by_lr(job):
    from fourm.utils.optim_factory import FusedAdamW
    model, batch = job
    lr = 1e-3
    opt = FusedAdamW([{"params": [p for p in model.parameters() if p.requires_grad], "weight_decay": 0.0}], lr=lr, betas=(0.9, 0.95), eps=1e-8)
    model.zero_grad(set_to_none=True)
    random.seed(3); loss, _ = model(batch, 128, 128); loss.backward()
    before = model.engine.flat_params.detach().clone()
    g = flat_grads(model)
    opt.step()
    delta = model.engine.flat_params.detach() - before
    assert float(delta.abs().max()) <= lr * 1.0001
    big = g.abs() > 1e-5                                             # update = -lr * g / (|g| + eps): ~ -lr * sign(g) there
    assert torch.equal(torch.sign(delta[big]), -torch.sign(g[big]))
    assert float((delta[big].abs() - lr).abs().max()) < 0.01 * lr
    assert float(delta[g == 0].abs().max()) == 0.0
