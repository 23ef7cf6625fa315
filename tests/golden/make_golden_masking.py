#!/usr/bin/env python3
"""Golden fixture for the device-side token budgets and sequence span masking from the UNMODIFIED upstream
``fourm.data.masking.UnifiedMasking`` (container only: needs /root/reference and its tokenizer file).

Upstream draws its random numbers inside the functions; here its samplers are patched to REPLAY recorded draws (Dirichlet.sample /
sample_n, torch.rand, random.uniform, np.random.randint), so that upstream, oracle/masking_oracle.py and csrc/masking.hip are functions
of the same numbers.  The script asserts oracle == upstream on every case and writes the draws, the inputs and upstream's outputs to
tests/golden/masking.npz.        python tests/golden/make_golden_masking.py [--check]"""
import argparse
import os
import random
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import ref_stubs  # noqa: E402

ref_stubs.install()
import fourm.data.masking as RM  # noqa: E402
from tokenizers import Tokenizer  # noqa: E402

from oracle import masking_oracle as MO  # noqa: E402

T_BUD, T_SEQ = 6, 32            # recorded tries per budget case / noise rows per sequence case
TOK = os.path.join(ref_stubs.REFERENCE_ROOT, "fourm/utils/tokenizer/trained/text_tokenizer_4m_wordpiece_30k.json")

WORDS = ("a photo of the cat sitting on a mat near two dogs and one red car in front of old house with green door under blue sky "
         "while people walk along river bank holding umbrellas during heavy rain").split()


def modality_info():
    mk = lambda typ, mx, mn, ia, ta, **kw: dict(type=typ, max_tokens=mx, min_tokens=mn, input_alphas=ia, target_alphas=ta, **kw)
    return {
        "rgb@224": mk("img", 196, 0, [1.0, 0.5], [1.0, 0.0]),
        "tok_depth@224": mk("img", 196, 0, [1.0, 0.5], [1.0, 1.0]),
        "caption": mk("seq", 256, 0, [1.0, 5.0], [1.0, 1.0], keep=["random", "all"]),
        "det": mk("seq", 256, 0, [1.0, 0.05], [1.0, 1.0], keep=["random", "binary"]),
        "tok_global": mk("seq_token", 16, 0, [0.5, 0.0], [0.5, 1.0], vocab_offset=300),
        "t5_caption": mk("seq_emb", 77, 0, [0.2, 5.0], [0.0, 0.0], keep=["random", "random"]),
    }


class Replay:
    """Patches the samplers UnifiedMasking uses and replays recorded draws."""

    def __init__(self):
        self.rows, self.kp, self.r, self.used = None, None, None, 0

    def rand(self, n, *a, **k):
        assert self.used < len(self.rows), "more retries than recorded noise rows"
        row = self.rows[self.used][:n]
        self.used += 1
        return torch.from_numpy(np.ascontiguousarray(row))

    def __enter__(self):
        self.saved = (torch.rand, random.uniform, random.choice, np.random.randint)
        torch.rand = self.rand
        random.uniform = lambda a, b: self.kp
        random.choice = lambda seq: self.kp
        np.random.randint = lambda n: self.r % n
        return self

    def __exit__(self, *exc):
        torch.rand, random.uniform, random.choice, np.random.randint = self.saved


def budget_cases(um, rng):
    M = um.num_modalities
    um.max_tries = T_BUD               # (upstream keeps the last try when none is valid: the recorded draws end there too)
    out = []
    for c in range(36):
        d = c % um.num_dirichlets
        n = int(rng.integers(8, 257))
        target = c % 3 == 2
        dist = (um.target_dirichlets if target else um.input_dirichlets)[d]
        conc = dist.concentration
        if c % 3 == 1:                     # force retries: a min_tokens not every draw meets
            um.min_tokens = torch.tensor([0, (0, 3, 8)[c % 3], (20, 40, 60, 90)[c % 4], 0, 0, 0])
        else:
            um.min_tokens = torch.zeros(M, dtype=torch.long)
        draw = lambda *shape: torch.distributions.Dirichlet(conc).sample(torch.Size(shape))
        torch.manual_seed(1000 + c)
        main = draw(T_BUD).float()
        extra = draw(T_BUD, M + 1).float()
        it = iter(range(T_BUD))
        state = {"t": -1}

        def sample():
            state["t"] = next(it)
            return main[state["t"]]

        dist.sample = sample
        dist.sample_n = lambda k: extra[state["t"], :int(k)]
        if target:
            ib = [int(v) for v in rng.integers(0, 120, M)]
            got = um.target_token_budget(ib, n, d)
            mx = MO.max_tokens_remaining(um.mod_is_img.numpy(), um.max_tokens.numpy(), um.min_tokens.numpy(), ib)
        else:
            ib = [0] * M
            got = um.input_token_budget(n, d)
            mx = um.max_tokens.numpy().astype(np.int32)
        del dist.sample, dist.sample_n
        b, tries = MO.token_budget(main.numpy(), extra.numpy(), n, um.min_tokens.numpy(), mx)
        assert list(b) == got, (c, list(b), got)
        assert min(tries, T_BUD) == state["t"] + 1          # (T_BUD + 1 = no draw met the minimum: upstream keeps the last one)
        out.append(dict(main=main.numpy(), extra=extra.numpy(), n=n, mn=um.min_tokens.numpy().astype(np.int32), mx=mx, out=np.array(got, dtype=np.int32),
                        tries=tries, target=target, in_budget=np.array(ib, dtype=np.int32)))
    um.min_tokens = torch.zeros(M, dtype=torch.long)
    return out


def seq_noise(c, width):
    """The noise rows of sequence case c (regenerated from the seed on the test side; the fixture keeps a checksum)."""
    return np.random.default_rng(5000 + c).random((T_SEQ, width), dtype=np.float32)


def sentence(rng, n):
    return " ".join(WORDS[int(i)] for i in rng.integers(0, len(WORDS), n))


def sequence_cases(um, tok, rng):
    out = []
    eos = um.eos_id
    with Replay() as rp:
        for c in range(40):
            kind = ("text", "chunks", "tokens")[c % 3]
            max_tokens = (256, 256, 16)[c % 3] if c % 7 else (24, 40, 8)[c % 3]
            keep = ("random", "all", "binary", "random")[c % 4]
            in_budget = int(rng.integers(0, min(60, max_tokens) + 1)) if c % 6 else 0        # (budgets never exceed max_tokens: clamp at :194)
            tgt_budget = None if c % 5 == 3 else int(rng.integers(0, min(50, max_tokens) + 1))
            rp.rows = seq_noise(c, 2 * max_tokens + 2)
            rp.used = 0
            rp.kp = {"random": float(rng.random()), "all": 1.0, "binary": float(rng.integers(0, 2))}[keep]
            rp.r = int(rng.integers(0, 1 << 20))
            unit, voff = None, 0
            if kind == "text":
                s = sentence(rng, int(rng.integers(1, 70)))
                ids = tok.encode(s).ids + [eos]
                ref = um.sequence_mask(s, max_tokens, in_budget, tgt_budget, keep)
            elif kind == "chunks":
                chunks = [sentence(rng, int(rng.integers(1, 6))) for _ in range(int(rng.integers(1, 30)))]
                enc = [e.ids for e in tok.encode_batch(chunks)] + [[eos]]
                ids = [t for ch in enc for t in ch]
                unit = [u for u, ch in enumerate(enc) for _ in ch]
                ref = um.sequence_mask(chunks, max_tokens, in_budget, tgt_budget, keep)
            else:
                voff = 300
                ids = [int(v) for v in rng.integers(0, 8192, int(rng.integers(1, max_tokens + 1)))]      # (seq_token: no truncation upstream)
                ref = um.sequence_token_mask(np.array(ids), max_tokens, in_budget, tgt_budget, keep, vocab_offset=voff)
            kp0 = 1.0 if keep == "all" else rp.kp
            o = MO.sequence_mask(ids, max_tokens, in_budget, tgt_budget, kp0, rp.rows, rp.r, um.sentinel_to_id, um.pad_id, unit_of=unit, vocab_offset=voff)
            for k in ("tensor", "input_mask", "target_mask", "decoder_attention_mask"):
                assert np.array_equal(o[k], ref[k].numpy()), (c, kind, k)
            assert o["tries"] == max(rp.used, 1), (c, o["tries"], rp.used)
            out.append(dict(ids=ids, unit=unit, max_tokens=max_tokens, in_budget=in_budget, tgt_budget=-1 if tgt_budget is None else tgt_budget, kp=kp0,
                            noise=rp.rows.copy(), r=rp.r, voff=voff, ref={k: ref[k].numpy() for k in ref}, tries=o["tries"]))
    return out


def emb_cases(um, rng):
    out = []
    with Replay() as rp:
        for c in range(8):
            n, D, max_tokens = (77, 8, 77) if c % 2 else (40, 8, 32)
            emb = rng.standard_normal((n, D), dtype=np.float32)
            in_budget = int(rng.integers(0, 50)) if c != 3 else 0
            keep = ("random", "all")[c % 2]
            rp.rows = seq_noise(100 + c, n)
            rp.used = 0
            rp.kp = float(rng.random())
            ref = um.sequence_emb_mask_span(torch.from_numpy(emb), max_tokens, in_budget, None, keep)
            kp0 = 1.0 if keep == "all" else rp.kp
            o = MO.sequence_emb_mask(emb, max_tokens, in_budget, kp0, rp.rows, um.sentinel_to_id)
            for k in ("tensor", "input_mask", "target_mask", "decoder_attention_mask"):
                assert np.array_equal(o[k], ref[k].numpy()), (c, k)
            out.append(dict(emb=emb, max_tokens=max_tokens, in_budget=in_budget, kp=kp0, noise=rp.rows.copy(), tensor=ref["tensor"].numpy(),
                            input_mask=ref["input_mask"].numpy()))
    return out


def pad_stack(rows, width, fill, dtype):
    a = np.full((len(rows), width), fill, dtype=dtype)
    for i, r in enumerate(rows):
        a[i, :len(r)] = r
    return a


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    tok = Tokenizer.from_file(TOK)
    um = RM.UnifiedMasking(modality_info(), tok, input_tokens_range=128, target_tokens_range=128)
    rng = np.random.default_rng(7)
    bud, seq, emb = budget_cases(um, rng), sequence_cases(um, tok, rng), emb_cases(um, rng)
    print(f"upstream == oracle: {len(bud)} budget cases (tries {sorted(set(b['tries'] for b in bud))}), {len(seq)} sequence cases "
          f"(tries up to {max(s['tries'] for s in seq)}, {sum(s['tgt_budget'] >= 0 and (~s['ref']['target_mask']).sum() == s['tgt_budget'] for s in seq)} truncated targets), "
          f"{len(emb)} embedding cases")
    if a.check:
        return
    sent = np.array([um.sentinel_to_id[k] for k in sorted(um.sentinel_to_id)], dtype=np.int32)
    assert sorted(um.sentinel_to_id) == list(range(len(sent)))
    W = max(len(s["ids"]) for s in seq)
    Lo = max(len(s["ref"]["tensor"]) for s in seq)
    fx = {
        "meta/sentinel_ids": sent, "meta/pad_id": np.array(um.pad_id), "meta/eos_id": np.array(um.eos_id),
        "bud/main": np.stack([b["main"] for b in bud]), "bud/extra": np.stack([b["extra"] for b in bud]), "bud/n": np.array([b["n"] for b in bud], dtype=np.int32),
        "bud/min": np.stack([b["mn"] for b in bud]), "bud/max": np.stack([b["mx"] for b in bud]), "bud/out": np.stack([b["out"] for b in bud]),
        "bud/tries": np.array([b["tries"] for b in bud], dtype=np.int32), "bud/is_target": np.array([b["target"] for b in bud]),
        "bud/in_budget": np.stack([b["in_budget"] for b in bud]), "bud/is_img": um.mod_is_img.numpy(), "bud/max_tokens": um.max_tokens.numpy().astype(np.int32),
        "seq/ids": pad_stack([s["ids"] for s in seq], W, 0, np.int32), "seq/len": np.array([len(s["ids"]) for s in seq], dtype=np.int32),
        "seq/unit": pad_stack([s["unit"] if s["unit"] is not None else [] for s in seq], W, -1, np.int32),
        "seq/chunked": np.array([s["unit"] is not None for s in seq]),
        "seq/max_tokens": np.array([s["max_tokens"] for s in seq], dtype=np.int32), "seq/in_budget": np.array([s["in_budget"] for s in seq], dtype=np.int32),
        "seq/tgt_budget": np.array([s["tgt_budget"] for s in seq], dtype=np.int32), "seq/kp": np.array([s["kp"] for s in seq], dtype=np.float64),
        "seq/noise_sum": np.array([float(s["noise"].astype(np.float64).sum()) for s in seq]), "seq/noise_width": np.array([s["noise"].shape[1] for s in seq], dtype=np.int32),
        "seq/r": np.array([s["r"] for s in seq], dtype=np.int32), "seq/voff": np.array([s["voff"] for s in seq], dtype=np.int32),
        "seq/tries": np.array([s["tries"] for s in seq], dtype=np.int32),
        "seq/out_len": np.array([len(s["ref"]["tensor"]) for s in seq], dtype=np.int32),
        "seq/tensor": pad_stack([s["ref"]["tensor"] for s in seq], Lo, 0, np.int32),
        "seq/input_mask": pad_stack([s["ref"]["input_mask"] for s in seq], Lo, True, bool),
        "seq/target_mask": pad_stack([s["ref"]["target_mask"] for s in seq], Lo, True, bool),
        "seq/dam": pad_stack([s["ref"]["decoder_attention_mask"] for s in seq], Lo, 0, np.int32),
    }
    for i, e in enumerate(emb):
        for k, v in e.items():
            fx[f"emb{i}/{k}"] = np.asarray(v) if k != "noise" else np.array(float(v.astype(np.float64).sum()))
    fx["meta/n_emb"] = np.array(len(emb)); fx["meta/t_seq"] = np.array(T_SEQ)
    np.savez_compressed(os.path.join(HERE, "masking.npz"), **fx)
    print("wrote masking.npz", os.path.getsize(os.path.join(HERE, "masking.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
