"""Generation step of 4M on MI355X: the MaskGIT decoding step of upstream ``GenerationSampler`` (fourm/models/generate.py:323-661).

One call of ``maskgit_step_batched`` = select + embed the conditioning tokens -> encoder -> context projection -> decoder over the
still-masked positions of the target modality (no self-attention mask, generate.py:642) -> logits -> temperature / top-k / top-p
sampling of every position -> keep the ``num_select`` most confident samples and write them into ``mod_dict``.  Everything runs on
the engine's HIP kernels; the sampling + commit are two launches of csrc/sample.hip instead of upstream's topk / sort / softmax /
cumsum / argsort / gather / multinomial / topk / scatter chain.

Scope: the MaskGIT and ROAR (random order) schemes, batched, with and without classifier-free guidance, for
grid-token target modalities, and autoregressive decoding of sequence modalities with a K/V cache (upstream re-runs the decoder
on the whole prefix per token, generate.py:850-914).  generate_sam_dense / generate_iter are not implemented and raise; nothing falls back to eager PyTorch.

Determinism: the only randomness is one uniform per decoded position drawn with ``torch.rand`` from the generator passed in (or the
device default): same logits + same uniforms -> same tokens (csrc/sample.hip, bit-exact against oracle/sample_oracle.py).
Upstream draws through ``torch.multinomial``; the sampled DISTRIBUTION is the same, the random stream is not.
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

__all__ = ["GenerationSampler"]


class GenerationSampler(nn.Module):
    def __init__(self, model):
        super().__init__()
        self.model = model

    # ------------------------------------------------------------------------------------------------------------------------
    # sampling  (generate.py:332-420)
    # ------------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _top_k_int(top_k, vocab):
        if not top_k:
            return 0
        if isinstance(top_k, int):
            return min(top_k, vocab)
        if isinstance(top_k, float):
            return min(int(top_k * vocab), vocab)
        raise ValueError(f"Invalid value for top_k: {top_k}")

    def sample_tokens(self, logits, temperature=1.0, top_k=0.0, top_p=0.0, generator=None, uniforms=None):
        """logits (R, V) bf16 / f32 -> (samples int64 (R), sampled_probs f32 (R)).  temperature ~ 0: argmax, probabilities 1."""
        from fourm.hip import _lib as L, ops
        if logits.dim() != 2 or logits.stride(1) != 1:
            raise ValueError("sample_tokens expects row-major (rows, vocab) logits")
        if logits.dtype not in (torch.bfloat16, torch.float32):
            logits = logits.float()
        R, V = logits.shape
        dev = logits.device
        greedy = abs(temperature) < 1e-10
        if uniforms is None and not greedy:
            uniforms = torch.rand(R, device=dev, generator=generator, dtype=torch.float32)
        ids = torch.empty(R, dtype=torch.int64, device=dev)
        probs = torch.empty(R, dtype=torch.float32, device=dev)
        L.check(L.sample_tokens(ops._p(logits), logits.stride(0), 1 if logits.dtype == torch.float32 else 0, R, V, float(temperature),
                                self._top_k_int(top_k, V), float(top_p or 0.0), ops._p(uniforms), ops._p(ids), ops._p(probs), ops._stream()))
        return ids, probs

    def sample_tokens_batched(self, logits, temperature=1.0, top_k=0.0, top_p=0.0, generator=None, uniforms=None):
        if logits.ndim > 2:
            B, N = logits.shape[:2]
            flat = logits.reshape(B * N, logits.shape[-1])
            s, p = self.sample_tokens(flat, temperature, top_k, top_p, generator, None if uniforms is None else uniforms.reshape(-1))
            return s.view(B, N), p.view(B, N)
        return self.sample_tokens(logits, temperature, top_k, top_p, generator, uniforms)

    # ------------------------------------------------------------------------------------------------------------------------
    # one MaskGIT forward  (generate.py:407-480, :628-647)
    # ------------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_enc_dec_maskgit_batched(self, mod_dict: Dict[str, Dict[str, torch.Tensor]], target_mod: str, seed: Optional[int] = None,
                                        decode_mask: Optional[torch.Tensor] = None):
        """-> (logits (B, N, V) for the N still-masked positions of ``target_mod`` in position order, mod_pos (B, N) int32).
        ``decode_mask`` (B, L) bool, False = decode this position: replaces the modality's target_mask for this forward (ROAR)."""
        from fourm.hip import _lib as L, ops
        m = self.model
        eng = m.engine
        eng.prepare()
        if target_mod not in m.decoder_embeddings:
            raise KeyError(f"{target_mod} has no decoder embedding")
        demb = m.decoder_embeddings[target_mod]
        if demb.kind != L.KIND_TOK:
            raise NotImplementedError("MaskGIT decoding is implemented for grid-token modalities (sequences use ROAR / autoregressive "
                                      "decoding upstream, which are not implemented here yet)")
        enc_names = [n for n in mod_dict if n in m.encoder_embeddings]
        B = mod_dict[enc_names[0]]["tensor"].shape[0]
        # budget = the largest number of visible tokens of a sample (generate.py:413, one host read like upstream's .max())
        vis = sum((~mod_dict[n]["input_mask"].reshape(B, -1).bool()).sum(1) for n in enc_names)
        n_enc = int(vis.max())
        tm = (mod_dict[target_mod]["target_mask"] if decode_mask is None else decode_mask).reshape(B, -1).bool()
        n_dec = int((~tm[0]).sum())                                  # "assumes num_decoder_tokens is the same across the batch" (:460)
        if n_enc == 0 or n_dec == 0:
            raise ValueError("nothing to condition on / nothing left to decode")
        enc = eng.select(mod_dict, n_enc, False, enc_names, "gen.enc.")
        dd = dict(mod_dict[target_mod])
        if decode_mask is not None:
            dd["target_mask"] = tm.view_as(dd["target_mask"]).contiguous()
        if "decoder_attention_mask" not in dd:                        # generation dicts need none (no decoder self-attention mask)
            dd["decoder_attention_mask"] = torch.zeros_like(tm, dtype=torch.int32)
        dec = eng.select({target_mod: dd}, n_dec, True, [target_mod], "gen.dec.", heads=[target_mod])
        dec["sa_mask"] = dict(mask_kind=L.MASK_NONE)
        dec.pop("cs")                                                 # trunk_forward: use the explicit (absent) self-attention mask
        y, _ = eng.trunk_forward(enc, dec, save=False)
        D, R = m.dim, B * n_dec
        yn = eng.ws.get("gen.yn", (y.shape[0], D), eng.adt)
        ops.layernorm_fwd(y, m.decoder_norm.weight, m.decoder_norm.bias, yn, eps=m.decoder_norm.eps, R=R)
        w = demb.to_logits.weight
        V = w.shape[0]
        lg = eng.ws.get("gen.logits", (y.shape[0], ops.ru(V, 8)), eng.adt)
        ops.gemm_nt(yn, eng.w(w), lg, M=R, N=V, K=D)
        return lg[:R, :V].view(B, n_dec, V) if lg.shape[1] == V else lg[:R].view(B, n_dec, -1)[..., :V], dec["slot_pos"].view(B, n_dec)

    @torch.no_grad()
    def maskgit_step_batched(self, mod_dict, target_mod, num_select, temperature, top_k, top_p, seed=None, generator=None, uniforms=None):
        """One MaskGIT step: mod_dict[target_mod]['tensor' | 'input_mask' | 'target_mask'] are updated in place (and returned)."""
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        logits, mod_pos = self.forward_enc_dec_maskgit_batched(mod_dict, target_mod)
        return self._sample_and_commit(mod_dict, target_mod, logits, mod_pos, num_select, temperature, top_k, top_p, generator, uniforms)

    def _sample_and_commit(self, mod_dict, target_mod, logits, mod_pos, num_select, temperature, top_k, top_p, generator, uniforms):
        """Sample every decoded position, keep the ``num_select`` most confident, write them into mod_dict (generate.py:393-405,
        :655-661): fm_sample_tokens + fm_maskgit_commit."""
        from fourm.hip import _lib as L, ops
        B, N, V = logits.shape
        rows = logits.reshape(B * N, V) if logits.is_contiguous() else logits.as_strided((B * N, V), (logits.stride(1), 1))
        samples, probs = self.sample_tokens(rows, temperature, top_k, top_p, generator, uniforms)
        d = mod_dict[target_mod]
        t = d["tensor"]
        flat = t.reshape(B, -1)
        if not flat.is_contiguous() or flat.data_ptr() != t.data_ptr():
            raise ValueError("mod_dict tensors must be contiguous (they are updated in place)")
        for k in ("input_mask", "target_mask"):
            if d[k].dtype != torch.bool or not d[k].is_contiguous():
                d[k] = d[k].bool().contiguous()
        num_select = min(int(num_select), N)
        top_idx = torch.empty(B, num_select, dtype=torch.int32, device=t.device)
        L.check(L.maskgit_commit(ops._p(probs), ops._p(samples), ops._p(mod_pos.contiguous()), B, N, num_select, ops._p(flat),
                                 1 if t.dtype == torch.int64 else 0, flat.shape[1], ops._p(d["input_mask"]), ops._p(d["target_mask"]),
                                 ops._p(top_idx), ops._stream()))
        self.last_step = dict(samples=samples.view(B, N), probs=probs.view(B, N), top_indices=top_idx, mod_pos=mod_pos)
        return mod_dict

    # ------------------------------------------------------------------------------------------------------------------------
    def generate_maskgit(self, mod_dict, target_mod, num_steps, temperature=1.0, top_k=0.0, top_p=0.0, generator=None):
        """A constant-rate MaskGIT schedule: ``num_steps`` steps, each committing an equal share of the remaining positions."""
        B = mod_dict[target_mod]["tensor"].shape[0]
        remaining = int((~mod_dict[target_mod]["target_mask"].reshape(B, -1)[0].bool()).sum())
        for step in range(num_steps):
            if remaining <= 0:
                break
            k = remaining if step == num_steps - 1 else max(1, remaining // (num_steps - step))
            self.maskgit_step_batched(mod_dict, target_mod, k, temperature, top_k, top_p, generator=generator)
            remaining -= k
        return mod_dict

    # ------------------------------------------------------------------------------------------------------------------------
    # ROAR = random order autoregression  (generate.py:481-514, :745-781)
    # ------------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def roar_decode_mask(self, mod_dict, target_mod, num_select, generator=None, order_noise=None):
        """The positions one ROAR step decodes: the ``num_select`` still-masked positions with the smallest entries of one random
        vector shared by the batch (upstream: argsort(target_mask + rand * 1e-6)[:, :num_select], generate.py:495-499).
        -> decode_mask (B, L) bool, False at the chosen positions."""
        tm = mod_dict[target_mod]["target_mask"]
        B = tm.shape[0]
        tm = tm.reshape(B, -1).bool()
        L_ = tm.shape[1]
        if order_noise is None:
            order_noise = torch.rand(L_, device=tm.device, generator=generator)
        n = min(int(num_select), int((~tm[0]).sum()))
        if n <= 0:
            raise ValueError("nothing left to decode")
        key = tm.float() + order_noise.reshape(1, L_).float() * 1e-6
        chosen = torch.argsort(key, dim=1)[:, :n]
        return torch.ones_like(tm).scatter_(1, chosen, False)

    @torch.no_grad()
    def forward_enc_dec_roar_batched(self, mod_dict, target_mod, num_select, seed=None, generator=None, order_noise=None):
        """-> (logits (B, n, V), mod_pos (B, n)) for the n = min(num_select, #masked) randomly chosen positions, in position order
        (upstream returns them in the random order; every position is sampled independently, so only the row order differs)."""
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        dm = self.roar_decode_mask(mod_dict, target_mod, num_select, generator, order_noise)
        return self.forward_enc_dec_maskgit_batched(mod_dict, target_mod, decode_mask=dm)

    @torch.no_grad()
    def roar_step_batched(self, mod_dict, target_mod, num_select, temperature, top_k, top_p, seed=None, generator=None, uniforms=None,
                          order_noise=None):
        """One ROAR step: every chosen position is sampled and committed (generate.py:766-781)."""
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        logits, mod_pos = self.forward_enc_dec_roar_batched(mod_dict, target_mod, num_select, generator=generator, order_noise=order_noise)
        return self._sample_and_commit(mod_dict, target_mod, logits, mod_pos, logits.shape[1], temperature, top_k, top_p, generator, uniforms)

    # ------------------------------------------------------------------------------------------------------------------------
    # classifier-free guidance  (generate.py:28-80, :665-703, :783-816)
    # ------------------------------------------------------------------------------------------------------------------------
    def unconditional_dict(self, mod_dict, conditioning):
        """A copy of ``mod_dict`` with the ``conditioning`` modalities emptied the way upstream's empty_img_modality /
        empty_seq_modality / empty_seq_emb_modality do (generate.py:30-80)."""
        out = {m: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()} for m, d in mod_dict.items()}
        for mod in conditioning:
            d, kind = out[mod], self.model.modality_info[mod]["type"]
            if kind in ("seq", "seq_token"):
                s1_id = 5                                      # id of the first sentinel token [S_1]
                d["tensor"][:] = 0
                d["tensor"][:, [0, 1]] = s1_id
                d["tensor"][:, -1] = s1_id + 1
                d["input_mask"][:] = True
                d["input_mask"][:, 0] = False
                d["target_mask"] = ~d["input_mask"]
                d["decoder_attention_mask"][:] = 1
                d["decoder_attention_mask"][:, 0] = 0
            elif kind == "seq_emb":
                d["tensor"] = torch.zeros_like(d["tensor"])
                d["input_mask"] = torch.ones_like(d["input_mask"])
                d["input_mask"][:, 0] = False
                d["target_mask"] = torch.ones_like(d["target_mask"])
                d["decoder_attention_mask"][:] = False
            else:
                d["input_mask"][:] = True
                d["target_mask"][:] = False
        return out

    @staticmethod
    def _combine(uncond, cond, weight, out=None):
        """out (f32, same leading shape) = (uncond | out) + weight * (cond - uncond): fm_guidance_combine on 2-D row views."""
        from fourm.hip import _lib as L, ops
        V = uncond.shape[-1]
        u2, c2 = uncond.reshape(-1, V), cond.reshape(-1, V)
        if u2.stride(1) != 1 or c2.stride(1) != 1:
            u2, c2 = u2.contiguous(), c2.contiguous()
        acc = out is not None
        if out is None:
            out = torch.empty(uncond.shape, dtype=torch.float32, device=uncond.device)
        L.check(L.guidance_combine(ops._p(u2), u2.stride(0), int(u2.dtype == torch.float32), ops._p(c2), c2.stride(0), int(c2.dtype == torch.float32),
                                   float(weight), ops._p(out), V, u2.shape[0], V, int(acc), ops._stream()))
        return out

    @torch.no_grad()
    def _guided_logits(self, mod_dict, target_mod, conditioning, guidance_scale, decode_mask=None):
        if target_mod in conditioning:
            raise ValueError("the target modality cannot be part of the conditioning that is dropped")
        cond, mod_pos = self.forward_enc_dec_maskgit_batched(mod_dict, target_mod, decode_mask=decode_mask)
        cond = cond.contiguous().clone()                      # (the next forward reuses the logits workspace)
        unc, _ = self.forward_enc_dec_maskgit_batched(self.unconditional_dict(mod_dict, conditioning), target_mod, decode_mask=decode_mask)
        return self._combine(unc, cond, guidance_scale), mod_pos

    @torch.no_grad()
    def guided_maskgit_step_batched(self, mod_dict, target_mod, num_select, temperature, top_k, top_p, conditioning=(), guidance_scale=1.0,
                                    seed=None, generator=None, uniforms=None, write_all_predictions=False):
        """MaskGIT step on logits_uncond + (logits_cond - logits_uncond) * guidance_scale (fp32), generate.py:665-703.
        ``write_all_predictions`` (generate_iter's visualisation mode, :696-697): every decoded position shows its current sample in
        ``tensor``; only the ``num_select`` most confident ones are committed in the masks."""
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        logits, mod_pos = self._guided_logits(mod_dict, target_mod, list(conditioning), guidance_scale)
        mod_dict = self._sample_and_commit(mod_dict, target_mod, logits, mod_pos, num_select, temperature, top_k, top_p, generator, uniforms)
        if write_all_predictions:
            t = mod_dict[target_mod]["tensor"]
            flat = t.reshape(t.shape[0], -1)
            # upstream's literal statement: tensor[:, mod_pos] = all_samples (for B > 1 every row receives the last row's samples)
            flat[:, self.last_step["mod_pos"].long()] = self.last_step["samples"].to(flat.dtype)
        return mod_dict

    @torch.no_grad()
    def guided_roar_step_batched(self, mod_dict, target_mod, num_select, temperature, top_k, top_p, conditioning=(), guidance_scale=1.0,
                                 seed=None, generator=None, uniforms=None, order_noise=None):
        """ROAR step with classifier-free guidance (generate.py:783-816): both passes decode the same random positions."""
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        dm = self.roar_decode_mask(mod_dict, target_mod, num_select, generator, order_noise)
        logits, mod_pos = self._guided_logits(mod_dict, target_mod, list(conditioning), guidance_scale, decode_mask=dm)
        return self._sample_and_commit(mod_dict, target_mod, logits, mod_pos, logits.shape[1], temperature, top_k, top_p, generator, uniforms)

    # conjunction of several weighted conditions: l_uncond + sum_i w_i (l_cond_i - l_uncond)   (generate.py:705-743, :817-848)
    @torch.no_grad()
    def _multi_guided_logits(self, uncond_dict, cond_dicts, cond_weights, target_mod, decode_mask=None):
        conds = []
        for cd in cond_dicts:
            lc, _ = self.forward_enc_dec_maskgit_batched(cd, target_mod, decode_mask=decode_mask)
            conds.append(lc.contiguous().clone())              # (copies: the next forward reuses the logits workspace)
        lu, mod_pos = self.forward_enc_dec_maskgit_batched(uncond_dict, target_mod, decode_mask=decode_mask)
        out = None
        for w, lc in zip(cond_weights, conds):
            out = self._combine(lu, lc, w, out)
        return out, mod_pos

    def _mirror_target(self, uncond_dict, cond_dicts, target_mod):
        for cd in cond_dicts:
            for k in ("tensor", "input_mask", "target_mask"):
                cd[target_mod][k] = uncond_dict[target_mod][k].clone()

    @torch.no_grad()
    def multi_guided_maskgit_step_batched(self, uncond_dict, cond_dicts, cond_weights, target_mod, num_select, temperature, top_k, top_p,
                                          seed=None, generator=None, uniforms=None):
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        logits, mod_pos = self._multi_guided_logits(uncond_dict, cond_dicts, cond_weights, target_mod)
        self._sample_and_commit(uncond_dict, target_mod, logits, mod_pos, num_select, temperature, top_k, top_p, generator, uniforms)
        self._mirror_target(uncond_dict, cond_dicts, target_mod)
        return uncond_dict, cond_dicts

    @torch.no_grad()
    def multi_guided_roar_step_batched(self, uncond_dict, cond_dicts, cond_weights, target_mod, num_select, temperature, top_k, top_p,
                                       seed=None, generator=None, uniforms=None, order_noise=None):
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        dm = self.roar_decode_mask(uncond_dict, target_mod, num_select, generator, order_noise)
        logits, mod_pos = self._multi_guided_logits(uncond_dict, cond_dicts, cond_weights, target_mod, decode_mask=dm)
        self._sample_and_commit(uncond_dict, target_mod, logits, mod_pos, logits.shape[1], temperature, top_k, top_p, generator, uniforms)
        self._mirror_target(uncond_dict, cond_dicts, target_mod)
        return uncond_dict, cond_dicts

    def generate_multi_guided(self, uncond_dict, cond_dicts, schedule, top_k=0.0, top_p=0.0, text_tokenizer=None, verbose=False, seed=None):
        """Chained generation under several weighted conditions (generate.py:1169-1228): a modality that has been generated becomes
        one more condition for the next one.  Image-like targets only, as upstream."""
        cp = lambda d: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        uncond_dict = {m: cp(d) for m, d in uncond_dict.items()}
        cond_dicts = [{m: cp(d) for m, d in cd.items()} for cd in cond_dicts]
        cur = schedule[0]["target_domain"]
        for cd in cond_dicts:
            cd[cur] = cp(uncond_dict[cur])
        for info in schedule:
            target, temp, k, weights = info["target_domain"], info["temperature"], info["num_tokens"], info["cfg_scale"]
            if cur != target:
                for cd in cond_dicts:
                    del cd[cur]
                    cd[target] = cp(uncond_dict[target])
                uncond_dict[cur]["input_mask"][:] = True
                new_cond = {cur: cp(uncond_dict[cur]), target: cp(uncond_dict[target])}
                new_cond[cur]["input_mask"][:] = False
                new_cond[cur]["target_mask"][:] = True
                cond_dicts.append(new_cond)
                cur = target
            if self.model.modality_info[target]["type"] != "img":
                raise NotImplementedError("Only image modalities are supported for now")
            scheme = info["scheme"].lower()
            if scheme == "maskgit":
                uncond_dict, cond_dicts = self.multi_guided_maskgit_step_batched(uncond_dict, cond_dicts, weights, target, k, temp, top_k, top_p, seed=seed)
            elif scheme == "roar":
                uncond_dict, cond_dicts = self.multi_guided_roar_step_batched(uncond_dict, cond_dicts, weights, target, k, temp, top_k, top_p, seed=seed)
            else:
                raise ValueError("Invalid sampling scheme")
        return uncond_dict

    # ------------------------------------------------------------------------------------------------------------------------
    # autoregressive decoding of a sequence modality with a K/V cache  (generate.py:516-548, :850-914)
    # ------------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def autoregressive_generate(self, mod_dict, target_mod, temperature=1.0, top_k=0.0, top_p=0.0, use_eos=True, eos_token=None,
                                start_tokens=None, generator=None, uniforms=None, keep_logits=False, conditioning=(), guidance_scale=1.0,
                                use_graphs=False):
        """Upstream ``autoregressive_step_batched`` / ``guided_autoregressive_step_batched`` up to (not including) the tokenizer-level
        merge: returns the generated ids ``out`` (B, T) starting with the start token(s).

        Upstream re-runs the whole decoder on the growing prefix for every token (no K/V cache, generate.py:885-896).  Here each
        layer keeps the prefix's q | k | v rows in a (B, T_max, 3 D) cache the qkv GEMM writes into directly, the new token attends
        to the filled part of it (fm_attn_fwd, kv_batch_rows = T_max), and the cross-attention keys / values of the encoded
        context are computed once per layer: one token costs one pass over B rows instead of B * prefix rows.
        With ``conditioning`` and ``guidance_scale != 1`` a second decoder state runs on the context of the emptied conditioning
        (classifier-free guidance, generate.py:919-1027) and the last-token logits are combined in fp32.
        ``uniforms`` (steps, B) fixes the random draws (one per sampled token).
        ``use_graphs``: one token is ~135 small launches (launch bound: 1.7 ms per token on 4M-B whatever the batch); with this flag
        the launch sequence of each sequence position is captured once into a hipGraph (kept on the sampler, keyed by position and
        shapes) and replayed afterwards - same kernels, same arguments, bit-identical logits."""
        from fourm.hip import _lib as L, ops
        m = self.model
        eng = m.engine
        eng.prepare()
        demb = m.decoder_embeddings[target_mod]
        if demb.kind != L.KIND_SEQ:
            raise NotImplementedError("autoregressive decoding is for sequence modalities (grid tokens: MaskGIT / ROAR steps)")
        if eng.qk_norm or eng.fp32:
            raise NotImplementedError("the K/V-cache decode path covers the bf16 models without qk_norm")
        D, H, ws, bf, f32 = eng.D, eng.H, eng.ws, eng.adt, torch.float32
        guided = len(conditioning) > 0 and float(guidance_scale) != 1.0
        if target_mod in conditioning:
            raise ValueError("the target modality cannot be part of the conditioning that is dropped")
        enc_names = [n for n in mod_dict if n in m.encoder_embeddings]
        B = mod_dict[enc_names[0]]["tensor"].shape[0]
        # ---- the target sequence: unmasked positions in order, their embeddings, start / end tokens (generate.py:516-548) ----
        d = demb.forward_embed(dict(mod_dict[target_mod]))
        tm = d["target_mask"].reshape(B, -1).bool()
        ids_all, emb_all = d["ids"].reshape(B, -1), d["emb"]
        n = int((~tm[0]).sum())                                      # "assumes num_decoder_tokens is the same across the batch"
        if n == 0:
            raise ValueError("no target positions")
        order = torch.argsort(tm.float() + torch.arange(tm.shape[1], device=tm.device).unsqueeze(0) * 1e-6, dim=1)[:, :n]
        dec_ids = torch.gather(ids_all, 1, order)
        dec_emb = torch.gather(emb_all, 1, order.unsqueeze(-1).expand(-1, -1, D))
        # samples with fewer target positions than sample 0 pull masked positions in: upstream zeroes their ids and embeddings
        # (forward_mask_decoder_autoregressive, generate.py:543-545)
        gathered_tm = torch.gather(tm, 1, order)
        dec_ids = dec_ids.masked_fill(gathered_tm, 0)
        dec_emb = dec_emb.masked_fill(gathered_tm.unsqueeze(-1), 0)
        T = min(int(m.modality_info[target_mod]["max_tokens"]), n)
        y_emb = ws.get("ar.y_emb", (B, T, D), f32)                   # (B, T, D): position + modality embedding of step t (a fixed
        y_emb.copy_(dec_emb[:, :T])                                  # buffer: captured graphs keep reading it in later calls)
        if use_eos and eos_token is None:
            eos_token = dec_ids[0][~torch.gather(tm, 1, order)[0]][-1]
        out = dec_ids[:, :1].long() if start_tokens is None else start_tokens.to(dec_ids.device).long()
        if use_eos:
            eos_token = torch.as_tensor(eos_token, device=out.device)
            if bool((out == eos_token).any(dim=-1).all()):
                return out
        Rp = ops.ru(B, 128)
        Tc = T + out.shape[1]                                        # cache capacity: prefix + up to T new tokens
        table = demb.token_emb.weight
        w_logits = demb.to_logits.weight
        V = w_logits.shape[0]

        def make_decoder(cond_dict, slot):
            """Encode ``cond_dict`` once; -> decode(tok (B,), p) = logits (B, V) bf16 predicting position p + 1, with this context's
            own per-layer K/V cache (workspace names carry ``slot``: the guided run keeps two of them alive)."""
            names = [k for k in cond_dict if k in m.encoder_embeddings]
            vis = sum((~cond_dict[k]["input_mask"].reshape(B, -1).bool()).sum(1) for k in names)
            n_enc = int(vis.max())
            if n_enc == 0:
                raise ValueError("nothing to condition on")
            enc = eng.select(cond_dict, n_enc, False, names, f"gen.enc{slot}.")
            _, ctx, emask, _ = eng.encode_context(enc)
            N, Rc, Rcp = n_enc, B * n_enc, ctx.shape[0]
            pre = f"ar{slot}."
            kvc, cache = [], []
            hc = ws.get(pre + "hc", (Rcp, D), bf)
            for l, blk in enumerate(m.decoder):                      # context keys / values once per layer, an empty q|k|v cache
                ops.layernorm_fwd(ctx, blk.context_norm.weight, blk.context_norm.bias, hc, eps=blk.context_norm.eps, R=Rc)
                kv = ws.get(f"{pre}kv{l}", (Rcp, 2 * D), bf)
                ops.gemm_nt(hc, eng.w(blk.cross_attn.kv.weight), kv, bias=blk.cross_attn.kv.bias, M=Rc, N=2 * D, K=D)
                kvc.append(kv)
                cache.append(ws.get(f"{pre}cache{l}", (B, Tc, 3 * D), bf))
            ya, yb = ws.get(pre + "ya", (Rp, D), f32), ws.get(pre + "yb", (Rp, D), f32)
            h, o = ws.get(pre + "h", (Rp, D), bf), ws.get(pre + "o", (Rp, D), bf)
            y1, y2 = ws.get(pre + "y1", (Rp, D), f32), ws.get(pre + "y2", (Rp, D), f32)
            q2 = ws.get(pre + "q2", (Rp, D), bf)
            lg = ws.get(pre + "logits", (Rp, ops.ru(V, 8)), bf)

            def decode(tok, p):
                y = ya
                y[:B] = table[tok] + y_emb[:, min(p, T - 1)]
                for l, blk in enumerate(m.decoder):
                    sa, xa, c = blk.self_attn, blk.cross_attn, cache[l]
                    ops.layernorm_fwd(y, blk.norm1.weight, blk.norm1.bias, h, eps=blk.norm1.eps, R=B)
                    row = c.view(B, Tc * 3 * D)[:, p * 3 * D:(p + 1) * 3 * D]              # this token's q | k | v inside the cache
                    ops.gemm_nt(h, eng.w(sa.qkv.weight), row, bias=sa.qkv.bias, M=B, N=3 * D, K=D)
                    flat = c.view(B * Tc, 3 * D)
                    ops.attn_fwd(row[:, :D], flat[:, D:2 * D], flat[:, 2 * D:], o, B, H, 1, p + 1, eng.scale, kv_batch_rows=Tc,
                                 zero_attn=getattr(blk.self_attn, "allow_zero_attn", False))
                    ops.gemm_nt(o, eng.w(sa.proj.weight), y1, epilogue=L.EPI_RESIDUAL, res=y, bias=sa.proj.bias, M=B, N=D, K=D)
                    ops.layernorm_fwd(y1, blk.query_norm.weight, blk.query_norm.bias, h, eps=blk.query_norm.eps, R=B)
                    ops.gemm_nt(h, eng.w(xa.q.weight), q2, bias=xa.q.bias, M=B, N=D, K=D)
                    ops.attn_fwd(q2, kvc[l][:, :D], kvc[l][:, D:], o, B, H, 1, N, eng.scale, zero_attn=getattr(blk.cross_attn, "allow_zero_attn", False), **emask)
                    ops.gemm_nt(o, eng.w(xa.proj.weight), y2, epilogue=L.EPI_RESIDUAL, res=y1, bias=xa.proj.bias, M=B, N=D, K=D)
                    ops.layernorm_fwd(y2, blk.norm2.weight, blk.norm2.bias, h, eps=blk.norm2.eps, R=B)
                    y = yb if y is ya else ya
                    eng._mlp_fwd(blk.mlp, h, y2, y, B, Rp, None, "ar")
                ops.layernorm_fwd(y, m.decoder_norm.weight, m.decoder_norm.bias, h, eps=m.decoder_norm.eps, R=B)
                ops.gemm_nt(h, eng.w(w_logits), lg, M=B, N=V, K=D)
                return lg[:B, :V]
            decode.n_ctx = N
            return decode

        decoders = [make_decoder(mod_dict, 0)]
        if guided:
            decoders.append(make_decoder(self.unconditional_dict(mod_dict, list(conditioning)), 1))

        def step_logits(tok, p):
            lc = decoders[0](tok, p)
            if not guided:
                return lc
            return self._combine(decoders[1](tok, p), lc, guidance_scale)

        if use_graphs:
            eager_step = step_logits
            tok_buf = ws.get("ar.tok", (B,), torch.int64)
            store = self.__dict__.setdefault("_ar_graphs", {})
            # everything that decides a workspace shape (hence a captured pointer) or a captured scalar
            shape_key = (id(m), target_mod, B, T, Tc, guided, float(guidance_scale), tuple(d.n_ctx for d in decoders))
            if store and next(iter(store))[0] != shape_key:          # graphs of another shape pin their memory pools: keep one shape
                store.clear()

            def step_logits(tok, p):                                 # noqa: F811  (replaces the eager step)
                tok_buf.copy_(tok)
                entry = store.get((shape_key, p))
                if entry is None:
                    eager_step(tok_buf, p)                           # allocations, shadow refreshes: outside the capture
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        res = eager_step(tok_buf, p)
                    entry = store[(shape_key, p)] = (g, res)
                entry[0].replay()
                return entry[1]

        for p in range(out.shape[1] - 1):                            # given start tokens beyond the first: fill the cache(s)
            for dec in decoders:
                dec(out[:, p], p)
        self.last_ar = dict(logits=[]) if keep_logits else None
        for i in range(T):
            p = out.shape[1] - 1
            logits = step_logits(out[:, p], p)
            if keep_logits:
                self.last_ar["logits"].append(logits.float().clone())
            u = None if uniforms is None else uniforms[i].contiguous()
            sample, _ = self.sample_tokens(logits, temperature, top_k, top_p, generator, u)
            out = torch.cat((out, sample.view(B, 1)), dim=-1)
            if use_eos and bool((out == eos_token).any(dim=-1).all()):
                break
        return out

    @staticmethod
    def sentinel_ids(text_tokenizer, match_str="[S_"):
        """Ids of the span-masking sentinel tokens "[S_0]", "[S_1]", ... of a ``tokenizers.Tokenizer`` (text_tokenizer.py:108-112)."""
        return {i for tok, i in text_tokenizer.get_vocab().items() if tok.startswith(match_str)}

    @staticmethod
    def merge_span_masking(input_ids, decoder_ids, sentinels):
        """Undo T5-style span masking (text_tokenizer.py:115-136): every sentinel of the input is replaced by the tokens the decoder
        produced after that sentinel; decoder tokens before its first sentinel are dropped."""
        spans, cur = {}, None
        for tok in decoder_ids:
            if tok in sentinels:
                cur = tok
                spans.setdefault(cur, [])
            elif cur is not None:
                spans[cur].append(tok)
        out = []
        for tok in input_ids:
            out.extend(spans.get(tok, [])) if tok in sentinels else out.append(tok)
        return out

    def merge_sequences_batched(self, mod_dict, pred_ids, target_mod, text_tokenizer, default_sentinel="[S_1]"):
        """Write generated sequences back into ``mod_dict[target_mod]`` (generate.py:550-626): per sample the visible input ids with
        their sentinels expanded by the prediction, right-padded with [PAD] to the longest sample; everything becomes input
        (input_mask False on real tokens), nothing is left to decode."""
        d = mod_dict[target_mod]
        dev = d["tensor"].device
        B = d["tensor"].shape[0]
        tens = d["tensor"].reshape(B, -1).cpu()
        vis = ~d["input_mask"].reshape(B, -1).bool().cpu()
        preds = pred_ids.reshape(B, -1).cpu().tolist()
        sent = self.sentinel_ids(text_tokenizer)
        pad_id = text_tokenizer.token_to_id("[PAD]")
        merged = []
        for b in range(B):
            inp = tens[b][vis[b]].tolist() or [text_tokenizer.get_vocab()[default_sentinel]]
            merged.append(self.merge_span_masking(inp, preds[b], sent))
        Lm = max(1, max(len(x) for x in merged))
        t = torch.full((B, Lm), pad_id, dtype=d["tensor"].dtype)
        im = torch.ones(B, Lm, dtype=torch.bool)
        for b, ids in enumerate(merged):
            t[b, :len(ids)] = torch.tensor(ids, dtype=t.dtype)
            im[b, :len(ids)] = False
        mod_dict[target_mod] = {"tensor": t.to(dev), "input_mask": im.to(dev), "target_mask": im.clone().to(dev),
                                "decoder_attention_mask": torch.zeros(B, Lm, dtype=torch.bool, device=dev)}
        return mod_dict

    def autoregressive_step_batched(self, mod_dict, target_mod, temperature, top_k, top_p, use_eos=True, eos_token=None, start_tokens=None,
                                    text_tokenizer=None, seed=None, generator=None):
        """Upstream's step (generate.py:850-914): decode the sequence (K/V cache) and merge it back into ``mod_dict``."""
        if text_tokenizer is None:
            raise ValueError("autoregressive_step_batched needs the text tokenizer to merge the prediction (sentinel / [PAD] ids); "
                             "autoregressive_generate() returns the raw ids without one")
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        out = self.autoregressive_generate(mod_dict, target_mod, temperature, top_k, top_p, use_eos, eos_token, start_tokens, generator)
        return self.merge_sequences_batched(mod_dict, out, target_mod, text_tokenizer)

    def guided_autoregressive_step_batched(self, mod_dict, target_mod, temperature, top_k, top_p, use_eos=True, eos_token=None,
                                           start_tokens=None, text_tokenizer=None, conditioning=(), guidance_scale=1.0, seed=None, generator=None):
        """Upstream's guided step (generate.py:919-1027): two decoder states (with / without the conditioning), fp32 combination."""
        if text_tokenizer is None:
            raise ValueError("guided_autoregressive_step_batched needs the text tokenizer to merge the prediction")
        if seed is not None:
            generator = torch.Generator(device=self.model.mask_token.device).manual_seed(seed)
        out = self.autoregressive_generate(mod_dict, target_mod, temperature, top_k, top_p, use_eos, eos_token, start_tokens, generator,
                                           conditioning=conditioning, guidance_scale=guidance_scale)
        return self.merge_sequences_batched(mod_dict, out, target_mod, text_tokenizer)

    # ------------------------------------------------------------------------------------------------------------------------
    # chained schedules  (generate.py:1029-1096)
    # ------------------------------------------------------------------------------------------------------------------------
    def _schedule_step(self, mod_dict, info, step, top_k, top_p, text_tokenizer, verbose, seed, show_all=False):
        target, temp = info["target_domain"], info["temperature"]
        scale, cond = info.get("cfg_scale", 1.0), list(info.get("cfg_cond_domains", []))
        seed_i = seed + step if seed is not None else None
        guided = scale != 1.0 and len(cond) > 0
        kind = self.model.modality_info[target]["type"]
        if verbose:
            print(f"[generate] step {step}: {target} {info.get('scheme', 'autoregressive')} temperature {temp}")
        if kind == "img":
            scheme, k = info["scheme"].lower(), info["num_tokens"]
            if scheme not in ("maskgit", "roar"):
                raise ValueError("Invalid sampling scheme")
            if guided and scheme == "maskgit":
                return self.guided_maskgit_step_batched(mod_dict, target, k, temp, top_k, top_p, conditioning=cond, guidance_scale=scale, seed=seed_i,
                                                        write_all_predictions=show_all)
            if guided:
                return self.guided_roar_step_batched(mod_dict, target, k, temp, top_k, top_p, conditioning=cond, guidance_scale=scale, seed=seed_i)
            fn = self.maskgit_step_batched if scheme == "maskgit" else self.roar_step_batched
            return fn(mod_dict, target, k, temp, top_k, top_p, seed=seed_i)
        if kind in ("seq", "seq_token"):
            if guided:
                return self.guided_autoregressive_step_batched(mod_dict, target, temp, top_k, top_p, text_tokenizer=text_tokenizer,
                                                               conditioning=cond, guidance_scale=scale, seed=seed_i)
            return self.autoregressive_step_batched(mod_dict, target, temp, top_k, top_p, text_tokenizer=text_tokenizer, seed=seed_i)
        raise ValueError("Invalid schedule")

    @staticmethod
    def _copy_mod_dict(mod_dict):
        return {m: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in d.items()} for m, d in mod_dict.items()}

    def generate(self, mod_dict, schedule, top_k=0.0, top_p=0.0, text_tokenizer=None, verbose=False, seed=None):
        """Run a generation schedule: a list of {target_domain, scheme, num_tokens, temperature, cfg_scale, cfg_cond_domains} steps
        (built by upstream's fourm/utils/generation.py build_chained_generation_schedules).  Works on a copy of ``mod_dict``."""
        mod_dict = self._copy_mod_dict(mod_dict)
        for step, info in enumerate(schedule):
            mod_dict = self._schedule_step(mod_dict, info, step, top_k, top_p, text_tokenizer, verbose, seed)
        return mod_dict

    @torch.no_grad()
    def generate_iter(self, mod_dict, schedule, top_k=0.0, top_p=0.0, text_tokenizer=None, verbose=False, seed=None):
        """``generate`` as an iterator (generate.py:1099-1161): yields the working mod_dict after every schedule step (the SAME dict object,
        updated in place, as upstream); guided MaskGIT steps show every current prediction in ``tensor`` (write_all_predictions)."""
        mod_dict = self._copy_mod_dict(mod_dict)
        for step, info in enumerate(schedule):
            mod_dict = self._schedule_step(mod_dict, info, step, top_k, top_p, text_tokenizer, verbose, seed, show_all=True)
            yield mod_dict

    @torch.no_grad()
    def generate_sam_dense(self, mod_dict, schedule, text_tokenizer, batch_size=16, key="sam_instance", top_k=0.0, top_p=0.0, seed=None,
                           verbose=False):
        """Dense SAM-instance prediction (generate.py:1230-1272): the single input is repeated ``batch_size`` times, the ``key`` sequence
        modality is generated on every copy (different samples), and the generated sequences are merged back at their sentinels and
        concatenated into ONE sequence that replaces ``mod_dict[key]``."""
        first = next(iter(mod_dict.values()))["tensor"]
        device = first.device
        mod_dict = self._copy_mod_dict(mod_dict)
        expanded = {}
        for m, d in mod_dict.items():                                   # expand_to_batch (utils/generation.py:185-195)
            expanded[m] = {}
            for k, v in d.items():
                if k in ("tensor", "input_mask", "target_mask", "decoder_attention_mask", "mask_valid") and torch.is_tensor(v):
                    if v.shape[0] == 1:
                        v = v.expand(batch_size, *v.shape[1:]).contiguous()
                    elif v.shape[0] != batch_size:
                        raise ValueError(f"Invalid batch size: {v.shape[0]} instead of {batch_size}")
                expanded[m][k] = v
        schedule = [s for s in schedule if s["target_domain"] == key]
        out = self.generate(expanded, schedule, text_tokenizer=text_tokenizer, verbose=verbose, seed=seed, top_p=top_p, top_k=top_k)
        sentinels = set(self.sentinel_ids(text_tokenizer))
        merged = []
        tens, im, tm = out[key]["tensor"].cpu(), out[key]["input_mask"].cpu(), out[key]["target_mask"].cpu()
        for i in range(batch_size):
            merged.extend(self.merge_span_masking(tens[i][im[i] == 0].tolist(), tens[i][tm[i] == 0].tolist(), sentinels))
        seq = torch.tensor(merged, device=device).unsqueeze(0)
        mod_dict[key] = {"tensor": seq, "input_mask": torch.zeros(seq.shape, dtype=torch.bool, device=device),
                         "target_mask": torch.ones(seq.shape, dtype=torch.bool, device=device),
                         "decoder_attention_mask": torch.zeros(seq.shape, dtype=torch.bool, device=device)}
        return mod_dict


# names only upstream's same-named module defines resolve lazily (see fourm/_upstream.py)
from fourm import _upstream as _up
__getattr__ = _up.fallthrough(__name__, is_package=False)
