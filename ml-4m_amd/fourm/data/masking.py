"""Device-side masking of a batch: what upstream's loader workers do per sample on the host
(``UnifiedMasking``, fourm/data/masking.py:131-564), as batched HIP kernels on tensors that are already on the GPU.

  image_mask_batched          UnifiedMasking.image_mask :236-266                     (fm_image_mask)
  token_budgets_batched       input_token_budget :181-205 / target_token_budget :207-234   (fm_token_budgets)
  sequence_mask_batched       sequence_mask :345-445 after tokenisation, sequence_token_mask :268-343   (fm_span_mask)
  sequence_emb_mask_batched   sequence_emb_mask_span :448-516                        (fm_span_mask, embedding mode)
  DeviceUnifiedMasking        UnifiedMasking.__call__ :519-564 over a whole batch: same constructor, same output contract

Every kernel is a pure function of (inputs, random draws): the draws are torch's on the device (or the caller's), consumed in the order
upstream consumes them, and the kernels are bit-exact against oracle/masking_oracle.py, which is pinned to the unmodified upstream
functions replaying the same draws (tests/golden/make_golden_masking.py).  Text tokenisation (the ``tokenizers`` WordPiece encoder) stays
host code: sequence modalities arrive as int32 ids with [EOS] appended, plus their lengths."""
from typing import Dict, Optional

import torch


@torch.no_grad()
def image_mask_batched(num_tokens: int, input_budget: torch.Tensor, target_budget: Optional[torch.Tensor] = None,
                       noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
    """input_budget / target_budget: int (B) device tensors (target_budget None = upstream's ``target_budget=None``).
    noise: optional f32 (B, num_tokens) uniform draws (default: torch.rand on the budgets' device)."""
    from fourm.hip import _lib as L, ops
    dev = input_budget.device
    B = input_budget.shape[0]
    if noise is None:
        noise = torch.rand(B, num_tokens, device=dev, generator=generator)
    noise = noise.float().contiguous()
    if tuple(noise.shape) != (B, num_tokens):
        raise ValueError(f"noise must be (B, num_tokens) = {(B, num_tokens)}")
    kin = input_budget.to(torch.int32).contiguous()
    kt = None if target_budget is None else target_budget.to(torch.int32).contiguous()
    im = torch.empty(B, num_tokens, dtype=torch.bool, device=dev)
    tm = torch.empty(B, num_tokens, dtype=torch.bool, device=dev)
    dam = torch.empty(B, num_tokens, dtype=torch.int32, device=dev)
    L.check(L.image_mask(ops._p(noise), ops._p(kin), ops._p(kt), B, num_tokens, ops._p(im), ops._p(tm), ops._p(dam), ops._stream()))
    return {"input_mask": im, "target_mask": tm, "decoder_attention_mask": dam}


def _i32(t, dev):
    return None if t is None else torch.as_tensor(t, device=dev).to(torch.int32).contiguous()


@torch.no_grad()
def token_budgets_batched(main_draws: torch.Tensor, extra_draws: torch.Tensor, num_tokens: torch.Tensor, min_tokens, max_tokens,
                          is_img=None, input_budget: Optional[torch.Tensor] = None):
    """main_draws f32 (B, T, M), extra_draws f32 (B, T, E, M): Dirichlet draws (see fm_token_budgets in include/fourm_hip.h).
    ``input_budget`` (B, M) switches to target budgets (the clamp of masking.py:218-219; needs ``is_img`` (M)).
    Returns (budget int32 (B, M), tries int32 (B))."""
    from fourm.hip import _lib as L, ops
    dev = main_draws.device
    B, T, M = main_draws.shape
    E = extra_draws.shape[2]
    if tuple(extra_draws.shape) != (B, T, E, M):
        raise ValueError(f"extra_draws must be (B, T, E, M) = {(B, T, E, M)}, got {tuple(extra_draws.shape)}")
    main_draws, extra_draws = main_draws.float().contiguous(), extra_draws.float().contiguous()
    n, mn, mx = _i32(num_tokens, dev), _i32(min_tokens, dev), _i32(max_tokens, dev)
    img = None if is_img is None else torch.as_tensor(is_img, device=dev).to(torch.uint8).contiguous()
    ib = _i32(input_budget, dev)
    if ib is not None and img is None:
        raise ValueError("target budgets need is_img")
    out = torch.empty(B, M, dtype=torch.int32, device=dev)
    tries = torch.empty(B, dtype=torch.int32, device=dev)
    L.check(L.token_budgets(ops._p(main_draws), ops._p(extra_draws), ops._p(n), ops._p(mn), ops._p(mx), ops._p(img), ops._p(ib), B, T, E, M,
                            ops._p(out), ops._p(tries), ops._stream()))
    return out, tries


def _span_common(a, B, dev, lengths, noise, keep_prob, input_budget, sentinel_ids, max_tokens):
    from fourm.hip import ops
    noise = noise.float().contiguous()
    if noise.dim() != 3 or noise.shape[0] != B:
        raise ValueError("noise must be (B, T, width)")
    keep = {"len": _i32(lengths, dev), "noise": noise, "kp": torch.as_tensor(keep_prob, device=dev).to(torch.float64).contiguous(),
            "ib": _i32(input_budget, dev), "sent": _i32(sentinel_ids, dev)}
    a.len, a.noise, a.keep_prob, a.input_budget, a.sentinel_ids = (ops._p(keep[k]) for k in ("len", "noise", "kp", "ib", "sent"))
    a.B, a.T, a.ld_noise, a.n_sentinels, a.max_tokens = B, noise.shape[1], noise.shape[2], keep["sent"].numel(), max_tokens
    return keep


@torch.no_grad()
def sequence_mask_batched(ids: torch.Tensor, lengths, max_tokens: int, input_budget, target_budget, keep_prob, noise: torch.Tensor,
                          sentinel_ids, pad_id: int, unit: Optional[torch.Tensor] = None, r_choice=None, vocab_offset: int = 0) -> Dict[str, torch.Tensor]:
    """Span masking of a batch of tokenised sequences (see fm_span_mask in include/fourm_hip.h for every argument).
    ids int (B, W) with ``lengths`` (B); noise f32 (B, T, >= number of mask decisions); keep_prob (B) f64; target_budget None = upstream's
    ``None``.  Returns tensor int32 / input_mask bool / target_mask bool / decoder_attention_mask int32, each (B, 2 * (max_tokens + 1)),
    plus ``tries`` (B) int32 (-1: more spans than sentinel ids, where upstream raises KeyError)."""
    from fourm.hip import _lib as L, ops
    dev = ids.device
    B, W = ids.shape
    a = L.SpanMaskArgs()
    keep = _span_common(a, B, dev, lengths, noise, keep_prob, input_budget, sentinel_ids, max_tokens)
    n_dec = min(W, max_tokens)
    if keep["noise"].shape[2] < n_dec:
        raise ValueError(f"noise rows hold {keep['noise'].shape[2]} draws, {n_dec} mask decisions are possible")
    ids32, unit32, tb, rc = _i32(ids, dev), _i32(unit, dev), _i32(target_budget, dev), _i32(r_choice, dev)
    Lo = 2 * (max_tokens + 1)
    out = {"tensor": torch.empty(B, Lo, dtype=torch.int32, device=dev), "input_mask": torch.empty(B, Lo, dtype=torch.bool, device=dev),
           "target_mask": torch.empty(B, Lo, dtype=torch.bool, device=dev), "decoder_attention_mask": torch.empty(B, Lo, dtype=torch.int32, device=dev),
           "tries": torch.empty(B, dtype=torch.int32, device=dev)}
    a.ids, a.unit, a.target_budget, a.r_choice = ops._p(ids32), ops._p(unit32), ops._p(tb), ops._p(rc)
    a.ld_ids, a.vocab_offset, a.pad_id = W, vocab_offset, pad_id
    a.tensor, a.input_mask, a.target_mask, a.decoder_attention_mask, a.tries = (ops._p(out[k]) for k in ("tensor", "input_mask", "target_mask", "decoder_attention_mask", "tries"))
    L.check(L.span_mask(a, ops._stream()))
    return out


@torch.no_grad()
def sequence_emb_mask_batched(emb: torch.Tensor, max_tokens: int, input_budget, keep_prob, noise: torch.Tensor, sentinel_ids,
                              lengths=None) -> Dict[str, torch.Tensor]:
    """``sequence_emb_mask_span`` over a batch: emb f32 (B, n, D).  Returns tensor f32 (B, max_tokens, D) (kept rows moved to the front in
    order, a zero row where a masked span starts), input_mask / target_mask bool, decoder_attention_mask int32 (B, max_tokens), tries."""
    from fourm.hip import _lib as L, ops
    dev = emb.device
    emb = emb.float().contiguous()
    B, n, D = emb.shape
    a = L.SpanMaskArgs()
    lengths = torch.full((B,), n, dtype=torch.int32, device=dev) if lengths is None else lengths
    keep = _span_common(a, B, dev, lengths, noise, keep_prob, input_budget, sentinel_ids, max_tokens)
    if keep["noise"].shape[2] < min(n, max_tokens):
        raise ValueError("noise rows too short")
    out = {"tensor": torch.empty(B, max_tokens, D, dtype=torch.float32, device=dev), "input_mask": torch.empty(B, max_tokens, dtype=torch.bool, device=dev),
           "target_mask": torch.empty(B, max_tokens, dtype=torch.bool, device=dev),
           "decoder_attention_mask": torch.empty(B, max_tokens, dtype=torch.int32, device=dev), "tries": torch.empty(B, dtype=torch.int32, device=dev)}
    src = torch.empty(B, max_tokens, dtype=torch.int32, device=dev)
    a.emb, a.emb_out, a.src, a.emb_rows, a.emb_dim = ops._p(emb), ops._p(out["tensor"]), ops._p(src), n, D
    a.input_mask, a.target_mask, a.decoder_attention_mask, a.tries = (ops._p(out[k]) for k in ("input_mask", "target_mask", "decoder_attention_mask", "tries"))
    L.check(L.span_mask(a, ops._stream()))
    return out


class DeviceUnifiedMasking:
    """``UnifiedMasking`` (fourm/data/masking.py:131-564) for a whole batch on the device: same constructor arguments, same per-modality
    output contract (tensor / input_mask / target_mask / decoder_attention_mask, here with a leading batch dimension - what the loader's
    collate produces from upstream's per-sample dicts).

    ``__call__(mod_dict)`` takes batched device tensors: image-like modalities as upstream ((B, ...) tensors, passed through), sequence
    modalities as ``{"ids": int (B, W), "len": int (B)[, "unit": int (B, W)]}`` (token ids with [EOS] appended: the host tokeniser's
    output; ``unit`` = chunk index per token for list-of-strings modalities), seq_token modalities as int (B, n) tensors, sequence embeddings
    as f32 (B, n, D).  The random numbers are torch's, drawn on the device from ``generator``; ``max_tries`` bounds both the budget tries
    (as upstream) and the keep-probability retries (upstream's loop is unbounded; see fm_span_mask)."""

    def __init__(self, modality_info: Dict, text_tokenizer, input_tokens_range, target_tokens_range, max_tries: int = 100,
                 sampling_weights=None, device="cuda", sentinel_to_id: Optional[Dict[int, int]] = None, pad_id: Optional[int] = None):
        two = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        self.input_tokens_range = two(input_tokens_range)
        self.target_tokens_range = two(target_tokens_range) if target_tokens_range is not None else None
        self.modality_info, self.num_modalities, self.max_tries = modality_info, len(modality_info), max_tries
        import os as _os
        self.check_every = int(_os.environ.get("FOURM_MASKING_CHECK_EVERY", "0"))      # batches between synchronising retry checks (0: only on .check())
        self.device = torch.device(device)
        info = list(modality_info.values())
        self.min_tokens = torch.tensor([m["min_tokens"] for m in info], dtype=torch.int32, device=self.device)
        self.max_tokens = torch.tensor([m["max_tokens"] for m in info], dtype=torch.int32, device=self.device)
        self.mod_is_img = torch.tensor([m["type"] == "img" for m in info], dtype=torch.uint8, device=self.device)
        eps = 1e-9                                                                      # (:160: alphas of 0 are clamped)
        self.input_alphas = torch.tensor([m["input_alphas"] for m in info], dtype=torch.float32, device=self.device).t().clamp(min=eps).contiguous()
        self.target_alphas = torch.tensor([m["target_alphas"] for m in info], dtype=torch.float32, device=self.device).t().clamp(min=eps).contiguous()
        if self.input_alphas.shape != self.target_alphas.shape:
            raise ValueError("input and target alphas describe different mixtures")
        self.num_dirichlets = self.input_alphas.shape[0]
        self.sampling_weights = None if sampling_weights is None else torch.tensor(sampling_weights, dtype=torch.float32, device=self.device)
        if self.sampling_weights is not None and len(sampling_weights) != self.num_dirichlets:
            raise ValueError("one sampling weight per Dirichlet mixture component")
        if sentinel_to_id is None:
            if text_tokenizer is None:
                raise ValueError("a text tokenizer (for its sentinel / [PAD] ids) or sentinel_to_id + pad_id")
            vocab = {k: v for k, v in text_tokenizer.get_vocab().items() if k.startswith("[S_")}          # text_tokenizer.py:108-112
            sentinel_to_id = {int(k.split("_")[1][:-1]): v for k, v in vocab.items()}
            pad_id = text_tokenizer.token_to_id("[PAD]")
        if sorted(sentinel_to_id) != list(range(len(sentinel_to_id))):
            raise ValueError("sentinel_to_id must map 0 .. n-1")
        self.sentinel_ids = torch.tensor([sentinel_to_id[k] for k in range(len(sentinel_to_id))], dtype=torch.int32, device=self.device)
        self.pad_id = int(pad_id)

    def _draws(self, alphas_b, T, E, g):
        """Dirichlet draws of every try: (B, T, M) and (B, T, E, M), as normalised Gamma(alpha, 1) variates (what Dirichlet.sample does)."""
        B, M = alphas_b.shape
        gam = torch._standard_gamma(alphas_b[:, None, None, :].expand(B, T, E + 1, M).contiguous(), generator=g)
        d = gam / gam.sum(-1, keepdim=True)
        return d[:, :, 0].contiguous(), d[:, :, 1:].contiguous()

    def _keep_prob(self, info, dir_idx, B, g):
        schemes = info.get("keep", ["random"] * self.num_dirichlets)
        u = torch.rand(B, device=self.device, generator=g, dtype=torch.float64)
        coin = torch.randint(0, 2, (B,), device=self.device, generator=g).to(torch.float64)
        kp = u.clone()
        for d, sch in enumerate(schemes):
            if sch == "all":
                kp = torch.where(dir_idx == d, torch.ones_like(kp), kp)
            elif sch == "binary":
                kp = torch.where(dir_idx == d, coin, kp)
            elif sch != "random":
                raise ValueError(f"Invalid keep scheme for sequence masking: {sch}")
        return kp

    @torch.no_grad()
    def __call__(self, mod_dict: Dict, generator: Optional[torch.Generator] = None, batch_size: Optional[int] = None) -> Dict[str, Dict[str, torch.Tensor]]:
        g, dev, M = generator, self.device, self.num_modalities
        first = next(iter(mod_dict.values()))
        B = batch_size or (first["ids"] if isinstance(first, dict) else first).shape[0]
        if self.sampling_weights is not None:
            dir_idx = torch.multinomial(self.sampling_weights, B, replacement=True, generator=g)
        else:
            dir_idx = torch.randint(0, self.num_dirichlets, (B,), device=dev, generator=g)
        n_in = torch.randint(self.input_tokens_range[0], self.input_tokens_range[1] + 1, (B,), device=dev, generator=g, dtype=torch.int32)
        T, E = self.max_tries, M
        main, extra = self._draws(self.input_alphas[dir_idx], T, E, g)
        in_budget, tr = token_budgets_batched(main, extra, n_in, self.min_tokens, self.max_tokens)
        self._note_tries(tr, T)
        tgt_budget = None
        if self.target_tokens_range is not None:
            n_tgt = torch.randint(self.target_tokens_range[0], self.target_tokens_range[1] + 1, (B,), device=dev, generator=g, dtype=torch.int32)
            main, extra = self._draws(self.target_alphas[dir_idx], T, E, g)
            tgt_budget, tr = token_budgets_batched(main, extra, n_tgt, self.min_tokens, self.max_tokens, is_img=self.mod_is_img, input_budget=in_budget)
            self._note_tries(tr, T)
        out = {}
        for m, (name, info) in enumerate(self.modality_info.items()):
            key = name if name in mod_dict else name.split("@")[0]                 # get_transform_key (modality_transforms.py:39-40)
            x = mod_dict[key]
            kin = in_budget[:, m].contiguous()
            kt = None if tgt_budget is None else tgt_budget[:, m].contiguous()
            typ, mt = info["type"], info["max_tokens"]
            if typ == "img":
                out[name] = dict(image_mask_batched(mt, kin, kt, generator=g), tensor=x)
                continue
            kp = self._keep_prob(info, dir_idx, B, g)
            tries = min(self.max_tries, 64)
            if typ == "seq_emb":
                noise = torch.rand(B, tries, min(x.shape[1], mt), device=dev, generator=g)
                out[name] = sequence_emb_mask_batched(x, mt, kin, kp, noise, self.sentinel_ids)
            elif typ in ("seq", "seq_token"):
                ids, lens, unit = (x["ids"], x["len"], x.get("unit")) if isinstance(x, dict) else (x, torch.full((B,), x.shape[1], dtype=torch.int32, device=dev), None)
                noise = torch.rand(B, tries, min(ids.shape[1], mt), device=dev, generator=g)
                r = torch.randint(0, 1 << 30, (B,), device=dev, generator=g, dtype=torch.int32)
                out[name] = sequence_mask_batched(ids, lens, mt, kin, kt, kp, noise, self.sentinel_ids, self.pad_id, unit=unit, r_choice=r,
                                                  vocab_offset=info.get("vocab_offset", 0) if typ == "seq_token" else 0)
            else:
                raise ValueError(f"Invalid modality type: {typ}")
            self._note_tries(out[name]["tries"], tries)
        self._calls = getattr(self, "_calls", 0) + 1
        every = getattr(self, "check_every", 0)
        if every and self._calls % every == 0:
            self.check()
        return out

    def _note_tries(self, tries: torch.Tensor, limit: int):
        """Device-side bookkeeping of the retry counters (no host synchronisation): how many samples ran out of sentinel ids (tries == -1,
        where upstream raises KeyError) and how many exhausted their retries (upstream's loops are unbounded and print 'More than max
        tries')."""
        st = getattr(self, "_tries_stat", None)
        if st is None or st.device != tries.device:
            st = self._tries_stat = torch.zeros(2, dtype=torch.int64, device=tries.device)
        st[0] += (tries < 0).sum()
        st[1] += (tries > limit).sum()                 # the kernels report limit + 1 when the draws ran out (limit = a fit on the last one)

    def check(self, raise_on_overflow: bool = True):
        """Synchronise once and report what the retry counters saw since the last check: raises on sentinel overflow (a batch with
        invalid sentinel ids was produced), warns about exhausted retries (a sample that may exceed its budget).  Called every
        ``check_every`` batches when that attribute is set (FOURM_MASKING_CHECK_EVERY), or by the trainer at epoch ends."""
        st = getattr(self, "_tries_stat", None)
        if st is None:
            return (0, 0)
        overflow, exhausted = (int(v) for v in st.tolist())
        st.zero_()
        if exhausted:
            # `force` is a keyword of the rank-aware print that fourm.utils.dist.setup_for_distributed installs; the stock print rejects it
            kw = {"force": True} if hasattr(print, "_fourm_builtin") else {}
            print(f"[DeviceUnifiedMasking] {exhausted} samples exhausted their retries (upstream: 'More than max tries')", **kw)
        if overflow and raise_on_overflow:
            raise KeyError(f"{overflow} samples needed more sentinel ids than the tokenizer provides (upstream raises KeyError in the masking transform)")
        return (overflow, exhausted)


# the host-side masking classes (UnifiedMasking, TransferMasking, ...) stay upstream's
from .. import _upstream as _up
_up.merge(__name__, globals())
