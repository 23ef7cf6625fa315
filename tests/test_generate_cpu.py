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


def test_generate_iter_and_sam_dense_host_logic(monkeypatch):
    """generate_iter yields after every step and asks guided MaskGIT steps to show all predictions; generate_sam_dense expands the batch,
    keeps only the key modality's schedule and merges the copies' sequences at their sentinels (generate.py:1099-1161, :1230-1272)."""
    from fourm.models.generate import GenerationSampler

    class M:
        modality_info = {"tok_depth@224": {"type": "img"}, "sam_instance": {"type": "seq"}}
    s = GenerationSampler(None)
    object.__setattr__(s, "model", M())
    calls = []

    def rec(name):
        def f(mod_dict, target, *a, **k):
            calls.append((name, target, k.get("write_all_predictions"), k.get("seed")))
            return mod_dict
        return f
    for n in ("maskgit_step_batched", "guided_maskgit_step_batched", "autoregressive_step_batched"):
        monkeypatch.setattr(s, n, rec(n))
    md = {"tok_depth@224": {"tensor": torch.zeros(1, 4)}, "sam_instance": {"tensor": torch.zeros(1, 6, dtype=torch.long)}}
    sched = [dict(target_domain="tok_depth@224", scheme="maskgit", num_tokens=2, temperature=1.0),
             dict(target_domain="tok_depth@224", scheme="maskgit", num_tokens=2, temperature=1.0, cfg_scale=3.0, cfg_cond_domains=["sam_instance"]),
             dict(target_domain="sam_instance", temperature=0.5)]
    seen = []
    for cur in s.generate_iter(md, sched, seed=7):
        seen.append(len(calls))
        assert cur is not md
    assert seen == [1, 2, 3]
    assert calls == [("maskgit_step_batched", "tok_depth@224", None, 7), ("guided_maskgit_step_batched", "tok_depth@224", True, 8),
                     ("autoregressive_step_batched", "sam_instance", None, 9)]
    # dense prediction: the generate() call sees a batch of 3 copies and only the key's schedule; its output is merged per copy
    tok = FakeTokenizer()
    S0, S1 = tok.vocab["[S_0]"], tok.vocab["[S_1]"]
    got = {}

    def fake_generate(mod_dict, schedule, **kw):
        got["batch"] = mod_dict["sam_instance"]["tensor"].shape[0]
        got["schedule"] = schedule
        out = {m: dict(d) for m, d in mod_dict.items()}
        t = torch.tensor([[20, S0, 21, S0, 30 + i, S1, 0] for i in range(3)])             # input "20 [S_0] 21", target "[S_0] x [S_1]"
        out["sam_instance"] = {"tensor": t, "input_mask": torch.tensor([[0, 0, 0, 1, 1, 1, 1]] * 3).bool(),
                               "target_mask": torch.tensor([[1, 1, 1, 0, 0, 0, 1]] * 3).bool()}
        return out
    monkeypatch.setattr(s, "generate", fake_generate)
    md2 = {"tok_depth@224": {"tensor": torch.arange(4.)[None], "input_mask": torch.zeros(1, 4).bool()},
           "sam_instance": {"tensor": torch.zeros(1, 7, dtype=torch.long), "input_mask": torch.ones(1, 7).bool(), "target_mask": torch.zeros(1, 7).bool()}}
    out = s.generate_sam_dense(md2, sched, tok, batch_size=3, key="sam_instance")
    assert got["batch"] == 3 and [x["target_domain"] for x in got["schedule"]] == ["sam_instance"]
    assert out["sam_instance"]["tensor"].tolist() == [[20, 30, 21, 20, 31, 21, 20, 32, 21]]
    assert not bool(out["sam_instance"]["input_mask"].any()) and bool(out["sam_instance"]["target_mask"].all())
    assert torch.equal(out["tok_depth@224"]["tensor"], md2["tok_depth@224"]["tensor"]) and out["tok_depth@224"]["tensor"] is not md2["tok_depth@224"]["tensor"]
    with pytest.raises(ValueError):
        bad = {"sam_instance": {"tensor": torch.zeros(2, 7, dtype=torch.long), "input_mask": torch.ones(2, 7).bool(), "target_mask": torch.zeros(2, 7).bool()}}
        s.generate_sam_dense(bad, sched, tok, batch_size=3, key="sam_instance")
