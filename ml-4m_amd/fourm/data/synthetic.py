"""Synthetic multimodal batches of the shapes the data loader hands to ``FourM.forward`` (SURVEY.md §8b,
§8d): random token ids, per-sample token budgets split over the modalities, generated directly on the
device.  Used by bench.py and __graft_entry__.smoke(); real batches come from the (CPU) data pipeline,
which is outside the hot path."""
import math
from typing import Dict, List

import torch


def modality_shapes(model, seq_tensor_len=None):
    """name -> dict(kind, L, ...) for every modality the model embeds, in sorted-name order (loader order)."""
    out = {}
    names = sorted(set(model.encoder_embeddings.keys()) | set(model.decoder_embeddings.keys()))
    for n in names:
        e = model.encoder_embeddings[n] if n in model.encoder_embeddings else model.decoder_embeddings[n]
        k = e.kind
        if k == 0:
            out[n] = dict(kind=k, L=e.num_patches, side=int(round(math.sqrt(e.num_patches))), vocab=e.vocab_size)
        elif k == 1:
            out[n] = dict(kind=k, L=e.num_patches, image=e.image_size, channels=e.num_channels)
        elif k == 2:
            L = seq_tensor_len or 2 * (e.max_length + 1)
            out[n] = dict(kind=k, L=L, vocab=e.vocab_size, room=min(e.max_length, L // 2))
        else:
            out[n] = dict(kind=k, L=e.max_length, dim=e.orig_emb_dim, room=e.max_length)
    return out


def _split_budget(gen, batch, caps: torch.Tensor, total: int, device):
    """Per-sample multinomial split of ``total`` over len(caps) bins, clipped to caps, remainder handed
    to bins with room (in order)."""
    n = caps.shape[-1]
    draws = torch.multinomial(torch.ones(batch, n, device=device), total, replacement=True, generator=gen)
    k = torch.zeros(batch, n, dtype=torch.long, device=device).scatter_add_(1, draws, torch.ones_like(draws))
    caps_b = caps[None].expand(batch, n) if caps.dim() == 1 else caps
    k = torch.minimum(k, caps_b)
    rest = total - k.sum(1)
    for i in range(n):
        add = torch.minimum(rest, caps_b[:, i] - k[:, i])
        k[:, i] += add
        rest -= add
    return k


@torch.no_grad()
def synthetic_batch(model, batch: int, n_in: int, n_out: int, device="cuda", seed: int = 0) -> Dict[str, Dict[str, torch.Tensor]]:
    gen = torch.Generator(device=device).manual_seed(seed)
    shp = modality_shapes(model)
    enc = [n for n in shp if n in model.encoder_embeddings]
    dec = [n for n in shp if n in model.decoder_embeddings]
    cap_in = torch.tensor([shp[n].get("room", shp[n]["L"]) for n in enc], device=device)
    k_in = _split_budget(gen, batch, cap_in, n_in, device)
    used = {n: k_in[:, i] for i, n in enumerate(enc)}
    cap_out = torch.stack([torch.full((batch,), shp[n].get("room", shp[n]["L"]), device=device)
                           - (used[n] if (shp[n]["kind"] in (0, 1) and n in used) else 0) for n in dec], 1)
    k_out = _split_budget(gen, batch, cap_out, n_out, device)
    got = {n: k_out[:, i] for i, n in enumerate(dec)}
    out = {}
    for n, s in shp.items():
        L, kind = s["L"], s["kind"]
        if kind == 0:
            t = torch.randint(0, s["vocab"], (batch, s["side"], s["side"]), device=device, generator=gen)
        elif kind == 1:
            t = torch.randn(batch, s["channels"], s["image"][0], s["image"][1], device=device, generator=gen)
        elif kind == 2:
            t = torch.randint(5, s["vocab"], (batch, L), device=device, generator=gen).int()
        else:
            t = torch.randn(batch, L, s["dim"], device=device, generator=gen)
        ki = used.get(n, torch.zeros(batch, dtype=torch.long, device=device))[:, None]
        ko = got.get(n, torch.zeros(batch, dtype=torch.long, device=device))[:, None]
        pos = torch.arange(L, device=device)[None]
        if kind in (2, 3):     # sequences: inputs first, then targets, attended causally
            rank = pos.expand(batch, L)
            dam = ((rank >= ki) & (rank < ki + ko)).int()
        else:                  # grids: a random permutation per sample, inputs first then targets
            rank = torch.rand(batch, L, device=device, generator=gen).argsort(1).argsort(1)
            tgt = (rank >= ki) & (rank < ki + ko)
            first = torch.where(tgt, pos.expand(batch, L), torch.full_like(rank, L)).min(1, keepdim=True).values
            dam = torch.where(pos == first, ko.expand(batch, L), torch.zeros_like(rank)).int()   # whole group attends to itself
        out[n] = dict(tensor=t, input_mask=~(rank < ki), target_mask=~((rank >= ki) & (rank < ki + ko)), decoder_attention_mask=dam)
    return out


def device_masking_for(model, num_input_tokens, num_target_tokens, device="cuda", alpha: float = 1.0, max_tries: int = 32):
    """A ``DeviceUnifiedMasking`` over the model's modalities (what upstream builds from the data YAML, run_training_4m.py:296-310): one
    Dirichlet component with concentration ``alpha`` on every modality the encoder / decoder embeds, span-masking sentinels 4 .. 203 as
    in the 4M text tokenizer."""
    from .masking import DeviceUnifiedMasking
    shp = modality_shapes(model)
    info = {}
    for n, s in shp.items():
        typ = {0: "img", 1: "img", 2: "seq", 3: "seq_emb"}[s["kind"]]
        info[n] = dict(type=typ, max_tokens=s.get("room", s["L"]) if typ != "img" else s["L"], min_tokens=0,
                       input_alphas=[alpha if n in model.encoder_embeddings else 0.0], target_alphas=[alpha if n in model.decoder_embeddings else 0.0])
    return DeviceUnifiedMasking(info, None, input_tokens_range=num_input_tokens, target_tokens_range=num_target_tokens, max_tries=max_tries, device=device,
                                sentinel_to_id={k: 4 + k for k in range(200)}, pad_id=0)


@torch.no_grad()
def device_masked_batch(model, masking, batch: int, device="cuda", generator=None) -> Dict[str, Dict[str, torch.Tensor]]:
    """Raw synthetic modalities (token grids, pixels, tokenised sequences with their lengths, sequence embeddings) pushed through the
    device-side masking pipeline (fourm.data.masking.DeviceUnifiedMasking): the mod_dict the loader would deliver, produced on the GPU."""
    shp = modality_shapes(model)
    raw = {}
    for n, s in shp.items():
        kind = s["kind"]
        if kind == 0:
            raw[n] = torch.randint(0, s["vocab"], (batch, s["side"], s["side"]), device=device, generator=generator)
        elif kind == 1:
            raw[n] = torch.randn(batch, s["channels"], s["image"][0], s["image"][1], device=device, generator=generator)
        elif kind == 2:
            room = s["room"]
            raw[n] = {"ids": torch.randint(204, s["vocab"], (batch, room), device=device, generator=generator, dtype=torch.int32),
                      "len": torch.randint(max(1, room // 8), room + 1, (batch,), device=device, generator=generator, dtype=torch.int32)}
        else:
            raw[n] = torch.randn(batch, s["L"], s["dim"], device=device, generator=generator)
    out = masking(raw, generator=generator, batch_size=batch)
    return {n: {k: v for k, v in d.items() if k != "tries"} for n, d in out.items()}


class SyntheticLoader:
    """An iterable with the training loader's output contract (one ``mod_dict`` per step, ``len()`` = steps per epoch): ``distinct``
    pre-generated synthetic batches handed out in turn.  Stands in for ``build_mixture_dataloader`` when the trainer runs without a
    dataset (``data_config`` of type 'synthetic', benchmarks, smoke tests)."""

    def __init__(self, model, batch_size: int, num_input_tokens: int, num_target_tokens: int, steps: int, device="cpu", seed: int = 0,
                 distinct: int = 4, masking: str = "uniform"):
        """``masking='uniform'``: pre-generated batches with uniform multinomial budgets (synthetic_batch).  ``'dirichlet'``: every step's
        batch is produced on the device by the masking kernels (Dirichlet budgets, image masks, span masking) from fresh draws."""
        self.steps = int(steps)
        self.live = None
        if masking == "dirichlet":
            if torch.device(device).type != "cuda":
                raise ValueError("device-side masking needs the GPU")
            gen = torch.Generator(device=device).manual_seed(seed)
            um = device_masking_for(model, num_input_tokens, num_target_tokens, device=device)
            self.live = lambda: device_masked_batch(model, um, batch_size, device=device, generator=gen)
            self.batches = []
            return
        if masking != "uniform":
            raise ValueError(f"masking {masking!r}: 'uniform' or 'dirichlet'")
        self.batches = [synthetic_batch(model, batch_size, num_input_tokens, num_target_tokens, device=device, seed=seed + i)
                        for i in range(max(1, min(distinct, self.steps)))]

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            b = self.live() if self.live is not None else self.batches[i % len(self.batches)]
            yield {m: dict(d) for m, d in b.items()}           # the forward adds keys to the inner dicts
