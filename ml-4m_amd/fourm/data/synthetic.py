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


class SyntheticLoader:
    """An iterable with the training loader's output contract (one ``mod_dict`` per step, ``len()`` = steps per epoch): ``distinct``
    pre-generated synthetic batches handed out in turn.  Stands in for ``build_mixture_dataloader`` when the trainer runs without a
    dataset (``data_config`` of type 'synthetic', benchmarks, smoke tests)."""

    def __init__(self, model, batch_size: int, num_input_tokens: int, num_target_tokens: int, steps: int, device="cpu", seed: int = 0,
                 distinct: int = 4):
        self.steps = int(steps)
        self.batches = [synthetic_batch(model, batch_size, num_input_tokens, num_target_tokens, device=device, seed=seed + i)
                        for i in range(max(1, min(distinct, self.steps)))]

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            b = self.batches[i % len(self.batches)]
            yield {m: dict(d) for m, d in b.items()}           # the forward adds keys to the inner dicts
