"""Host-side pieces of the generation API (no GPU): span-masking merge of generated sequences and the schedule dispatcher.
The merge is pinned against upstream's own helpers when the reference tree is present (build container only)."""
import importlib.util
import os
import random

import pytest
import torch

REF = "/root/reference/fourm/utils/tokenizer/text_tokenizer.py"


class FakeTokenizer:
    def __init__(self):
        self.vocab = {"[PAD]": 0, "[SOS]": 1, "[EOS]": 2, **{f"[S_{i}]": 5 + i for i in range(8)}, **{f"w{i}": 20 + i for i in range(50)}}

    def get_vocab(self):
        return dict(self.vocab)

    def token_to_id(self, tok):
        return self.vocab.get(tok)


def sampler():
    from fourm.models.generate import GenerationSampler
    return GenerationSampler(None)


def test_merge_span_masking_examples():
    s, tok = sampler(), FakeTokenizer()
    sent = s.sentinel_ids(tok)
    assert sent == set(range(5, 13))
    # "a [S_0] b [S_1]" + "[S_0] x y [S_1] z [S_2]"  ->  a x y b z
    assert s.merge_span_masking([20, 5, 21, 6], [5, 30, 31, 6, 32, 7], sent) == [20, 30, 31, 21, 32]
    assert s.merge_span_masking([20, 5], [40, 41, 5, 30], sent) == [20, 30]              # tokens before the first sentinel are dropped
    assert s.merge_span_masking([20, 9, 21], [5, 30], sent) == [20, 21]                  # a sentinel the decoder never produced
    assert s.merge_span_masking([5], [5, 30, 5, 31], sent) == [30, 31]                   # a repeated sentinel accumulates


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not on this machine")
def test_merge_matches_upstream_helpers():
    spec = importlib.util.spec_from_file_location("ref_text_tokenizer", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    s, tok = sampler(), FakeTokenizer()
    sent = s.sentinel_ids(tok)
    assert sent == set(ref.get_sentinel_to_id_mapping(tok).values())
    rng = random.Random(0)
    for _ in range(200):
        inp = [rng.choice(list(range(5, 10)) + list(range(20, 40))) for _ in range(rng.randint(1, 12))]
        dec = [rng.choice(list(range(5, 10)) + list(range(20, 40))) for _ in range(rng.randint(0, 16))]
        assert s.merge_span_masking(inp, dec, sent) == ref.merge_span_masking(inp, dec, sent)


def test_merge_sequences_batched_layout():
    s, tok = sampler(), FakeTokenizer()
    t = torch.tensor([[20, 5, 21, 6, 0, 0], [22, 5, 0, 0, 0, 0]])
    im = torch.tensor([[0, 0, 0, 0, 1, 1], [0, 0, 1, 1, 1, 1]], dtype=torch.bool)
    md = {"caption": {"tensor": t, "input_mask": im, "target_mask": ~im, "decoder_attention_mask": torch.zeros_like(im)}}
    pred = torch.tensor([[5, 30, 31, 6, 32, 2], [5, 33, 2, 0, 0, 0]])
    out = s.merge_sequences_batched(md, pred, "caption", tok)["caption"]
    # (EOS / PAD are ordinary tokens for the merge, exactly as upstream: they follow the last sentinel's span)
    assert out["tensor"].tolist() == [[20, 30, 31, 21, 32, 2], [22, 33, 2, 0, 0, 0]]
    assert out["input_mask"].tolist() == [[False] * 6, [False] * 6]
    assert torch.equal(out["target_mask"], out["input_mask"]) and not bool(out["decoder_attention_mask"].any())
    # ragged lengths are right-padded with [PAD] and masked
    out = s.merge_sequences_batched({"caption": {"tensor": t.clone(), "input_mask": im.clone()}}, torch.tensor([[5, 30, 6, 31], [5, 33, 34, 35]]),
                                    "caption", tok)["caption"]
    assert out["tensor"].tolist() == [[20, 30, 21, 31], [22, 33, 34, 35]]
    out = s.merge_sequences_batched({"caption": {"tensor": t.clone(), "input_mask": im.clone()}}, torch.tensor([[5, 30, 6, 31], [6, 33, 34, 35]]),
                                    "caption", tok)["caption"]
    assert out["tensor"].tolist() == [[20, 30, 21, 31], [22, 0, 0, 0]] and out["input_mask"].tolist() == [[False] * 4, [False, True, True, True]]


def test_generate_dispatches_the_schedule(monkeypatch):
    from fourm.models.generate import GenerationSampler

    class M:
        modality_info = {"tok_depth@224": {"type": "img"}, "caption": {"type": "seq"}}
    s = GenerationSampler(None)
    object.__setattr__(s, "model", M())
    calls = []

    def rec(name):
        def f(mod_dict, target, *a, **k):
            calls.append((name, target, a, {x: k[x] for x in sorted(k) if x != "text_tokenizer"}))
            return mod_dict
        return f
    for n in ("maskgit_step_batched", "guided_maskgit_step_batched", "roar_step_batched", "guided_roar_step_batched", "autoregressive_step_batched"):
        monkeypatch.setattr(s, n, rec(n))
    md = {"tok_depth@224": {"tensor": torch.zeros(1, 4)}, "caption": {"tensor": torch.zeros(1, 4)}}
    sched = [dict(target_domain="tok_depth@224", scheme="maskgit", num_tokens=7, temperature=1.5),
             dict(target_domain="tok_depth@224", scheme="ROAR", num_tokens=3, temperature=0.5, cfg_scale=2.0, cfg_cond_domains=["caption"]),
             dict(target_domain="tok_depth@224", scheme="roar", num_tokens=3, temperature=0.5, cfg_scale=2.0, cfg_cond_domains=[]),
             dict(target_domain="caption", scheme="autoregressive", num_tokens=None, temperature=0.7)]
    out = s.generate(md, sched, top_k=5, top_p=0.9, text_tokenizer="tok", seed=10)
    assert out is not md and out["caption"]["tensor"] is not md["caption"]["tensor"]
    assert [c[0] for c in calls] == ["maskgit_step_batched", "guided_roar_step_batched", "roar_step_batched", "autoregressive_step_batched"]
    assert calls[0][2] == (7, 1.5, 5, 0.9) and calls[0][3] == {"seed": 10}
    assert calls[1][3] == {"conditioning": ["caption"], "guidance_scale": 2.0, "seed": 11}
    assert calls[3][2] == (0.7, 5, 0.9) and calls[3][3] == {"seed": 13}
    with pytest.raises(ValueError):
        s.generate(md, [dict(target_domain="tok_depth@224", scheme="beam", num_tokens=1, temperature=1.0)])
