"""The sampling oracle against upstream's filter semantics and basic statistics.  CPU only."""
import numpy as np
import pytest

from oracle import sample_oracle as S


@pytest.mark.parametrize("V,top_k,top_p", [(64, 0, 0.9), (4096, 50, 0.0), (4096, 200, 0.8), (30000, 0, 0.5), (1000, 10, 0.99)])
def test_survivors_match_upstream_filter(V, top_k, top_p):
    rng = np.random.RandomState(V + top_k)
    for rep in range(4):
        row = (rng.randn(V) * 3).astype(np.float32)
        mine = S.survivors(row, top_k, top_p)
        ref = S.upstream_filter_survivors(row, top_k, top_p)
        # fp32 cumulative softmax vs 2^-24-quantised integer masses: the cut may move by entries whose cumulative mass sits within
        # the quantisation of top_p; never by more than a handful at the margin, and the bulk is identical
        diff = np.nonzero(mine != ref)[0]
        assert diff.size <= max(2, int(2e-3 * ref.sum())), (diff.size, ref.sum())
        if diff.size:
            p = np.exp(row - row.max()); p /= p.sum()
            assert p[diff].max() < 1e-3                         # only entries of negligible mass at the very edge of the nucleus


def test_exp_det_accuracy_and_greedy():
    x = -np.abs(np.random.RandomState(0).randn(10000).astype(np.float32)) * 12
    e = S.exp_det(x)
    assert np.max(np.abs(e - np.exp(x.astype(np.float64))) / np.exp(x.astype(np.float64))) < 4e-7
    lg = np.random.RandomState(1).randn(5, 100).astype(np.float32)
    ids, pr = S.sample_tokens(lg, 0.0, 0, 0.0, np.zeros(5, np.float32))
    assert np.array_equal(ids, lg.argmax(1)) and np.all(pr == 1)


def test_multinomial_follows_the_distribution():
    rng = np.random.RandomState(2)
    row = rng.randn(50).astype(np.float32)
    R = 4000
    ids, pr = S.sample_tokens(np.repeat(row[None], R, 0), 1.0, 0, 0.0, rng.rand(R).astype(np.float32))
    p = np.exp(row - row.max()); p /= p.sum()
    freq = np.bincount(ids, minlength=50) / R
    assert np.abs(freq - p).max() < 0.03
    assert np.allclose(pr, p[ids], rtol=1e-5)


def test_commit_keeps_the_most_confident():
    prob = np.array([[0.1, 0.9, 0.9, 0.3]], np.float32)
    samples = np.array([[7, 8, 9, 10]], np.int64)
    pos = np.array([[5, 2, 0, 3]], np.int32)
    t, im, tm = np.zeros((1, 6), np.int64), np.ones((1, 6), bool), np.zeros((1, 6), bool)
    top = S.maskgit_commit(prob, samples, pos, 2, t, im, tm)
    assert top.tolist() == [[1, 2]] and t[0, 2] == 8 and t[0, 0] == 9 and not im[0, 2] and tm[0, 0] and im[0, 5]


def test_roar_positions_follow_upstream_argsort():
    """oracle roar_positions == upstream's argsort(target_mask + rand * 1e-6)[:, :n] (generate.py:495-499), as a set per sample."""
    import torch
    g = torch.Generator().manual_seed(0)
    B, L_ = 3, 50
    tm = torch.rand(B, L_, generator=g) < 0.4
    tm[1] = tm[0]; tm[2] = tm[0]                      # upstream assumes one decoded count for the batch
    noise = torch.rand(L_, generator=g)
    n = 7
    ids = torch.argsort(tm + noise.unsqueeze(0) * 1e-6, dim=1)[:, :n]
    got = S.roar_positions(tm.numpy(), noise.numpy(), n)
    for b in range(B):
        assert sorted(ids[b].tolist()) == got[b].tolist()
        assert not tm[b, got[b]].any()
    assert S.roar_positions(tm.numpy(), noise.numpy(), 10 ** 6).shape[1] == int((~tm[0]).sum())
