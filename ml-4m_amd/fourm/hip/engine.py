"""Host-side executor of the 4M train step on one MI355X.

Owns (a) the flat fp32 parameter / gradient stores, (b) the bf16 weight shadows in the layouts the
GEMM kernels want (plain for y = x W^T, transposed for dX = dY W, both zero padded to multiples of
64 on the reduction axis), (c) the static activation workspace, and (d) the launch sequence of the
forward and of the hand-written backward.  There is no autograd graph inside the step: one
``torch.autograd.Function`` (in fourm/models/fm.py) hands the upstream gradient of the loss to
``train_backward`` which fills ``param.grad`` views of the flat gradient store.

Every shape in a step is static and nothing synchronises with the host: ragged per-modality row
counts are handled on the device (segment tables), so the whole step can be captured in a hipGraph.

Numerics follow CUDA autocast(bf16) as used upstream (run_training_4m.py:723): GEMM operands and
outputs bf16 with fp32 accumulation, LayerNorm / softmax / cross-entropy in fp32, fp32 residual
stream, fp32 master weights and gradients.
"""
import math
import os
import random
import weakref
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ops

_WEIGHT_EPOCH = 0
# run the activation backward inside the fc2 dX GEMM epilogue (bit-identical; measured slightly slower, see _mlp_bwd)
FUSE_ACT_BWD = os.environ.get("FOURM_FUSE_ACT_BWD", "0") == "1"
# Logits of the per-modality heads: one dense gemm_nt3 launch per head (row range from device memory) instead of the grouped kernel of
# gemm.hip, for up to HEADS_DENSE_MAX heads (each launch has a fixed cost: 7 heads of 4M-B mod7 -0.4 ms per step, the 21 heads of mod21 no gain
# in situ - profiles/r05_heads_dense.txt; FOURM_HEADS_DENSE=0: grouped always).
HEADS_DENSE = os.environ.get("FOURM_HEADS_DENSE", "1") == "1"
HEADS_DENSE_MAX = int(os.environ.get("FOURM_HEADS_DENSE_MAX", "8"))
# The residual add behind attn.proj / cross_attn.proj / mlp.fc2 runs in the LayerNorm that follows (fm_layernorm_fwd_res) instead of the
# GEMM epilogue: the GEMM becomes a plain bf16 launch (the lock-step kernel), the fp32 read-modify-write of the stream moves from an
# epilogue all workgroups enter together (~2.8 TB/s) into a streaming kernel (5.5 TB/s).  Bit-identical.  FOURM_DEFER_RESIDUAL=0: fused.
DEFER_RESIDUAL = os.environ.get("FOURM_DEFER_RESIDUAL", "1") == "1"
# Every decoder block normalises the SAME context (fm_utils.py:364, fm.py:514-515): x_hat = (context - mean) * rstd does not depend on the
# layer.  With bias-free context norms the engine computes x_hat once, keeps gamma_l inside the bf16 image of cross_attn.kv.weight
# (kv_l = x_hat (W_l diag(gamma_l))^T), sums the layers' gradients with respect to x_hat in ONE long-K GEMM and runs the LayerNorm backward
# once (FourMEngine.hoist_ctx).  Only the bf16 rounding point moves: bf16(x_hat gamma) bf16(W) -> bf16(x_hat) bf16(W gamma).
# FOURM_HOIST_CTX=0: one LayerNorm forward / backward per decoder block, as upstream computes it.
HOIST_CTX = os.environ.get("FOURM_HOIST_CTX", "1") == "1"
# LayerNorm backward of bias-free norms: x_hat = h / gamma from the saved bf16 norm output h (kept for the dW GEMM of the Linear it feeds)
# instead of (x - mean) * rstd from the fp32 input: 14 instead of 16 bytes per element of an HBM-bound kernel (fm_layernorm_bwd_h).
# x_hat then carries h's bf16 rounding (2^-9 relative - the operand the forward GEMMs used).  FOURM_LN_BWD_FROM_H=0: from x, as upstream's fp32 norm.
LN_BWD_FROM_H = os.environ.get("FOURM_LN_BWD_FROM_H", "1") == "1"
# lab: keep the data-parallel CU reservation for the whole step (the behaviour up to round 5) instead of the backward only
DP_RESERVE_ALWAYS = os.environ.get("FOURM_DP_RESERVE_ALWAYS", "0") == "1"


def bump_weight_epoch():
    """Called by optimizers that update parameters behind torch's version counters (FusedAdamW)."""
    global _WEIGHT_EPOCH
    _WEIGHT_EPOCH += 1


def ru(x, m):
    return (x + m - 1) // m * m


class Shadow:
    """One bf16 weight shadow: ``buf`` plus the jobs (fp32 master view -> bf16 view, transpose) that fill it and
    the stamp of the masters it was last built from."""
    __slots__ = ("buf", "params", "jobs", "stamp")

    def __init__(self, buf, params, jobs):
        self.buf, self.params, self.jobs, self.stamp = buf, params, jobs, None


class Workspace:
    """Named device buffers, allocated (zero-filled) on first use and reused while the shape matches.
    Padding rows / columns are never written by any kernel, so they stay zero for the GEMM contracts."""

    def __init__(self, device, per_stream: bool = False):
        """per_stream: a buffer set per HIP stream (the key carries the current stream's handle), so that calls issued on different streams - two
        tokenizer sub-batches in flight (fourm.vq.tokenize_sub_batches) - never share scratch.  Off for the train step (one stream, ~1500 lookups)."""
        self.device = device
        self.per_stream = per_stream
        self.bufs: Dict[tuple, torch.Tensor] = {}

    def get(self, name, shape, dtype):
        key = (name, tuple(shape), dtype, torch.cuda.current_stream(self.device).cuda_stream) if self.per_stream else (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            t = self.bufs[key] = torch.zeros(shape, dtype=dtype, device=self.device)
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


def fill_mod_desc(md, d, emb, is_dec: bool, mod_id: int, head_index: int = 0, raw: int = 0, name: str = "modality"):
    """One modality of ``mod_dict`` -> ``fm_mod_desc``.  Returns the tensors that must stay alive until the launch.
    raw == 2 (the embedding's own forward): no teacher-forcing shift, no decoder_attention_mask needed."""
    kind = emb.kind
    t = d["tensor"]
    B = t.shape[0]
    mask = d["target_mask" if is_dec else "input_mask"]
    mask = mask.reshape(B, -1)
    if mask.dtype != torch.bool:
        mask = mask.bool()
    mask = mask.contiguous()
    keep = [mask]
    md.mask, md.mask_stride = mask.data_ptr(), mask.shape[1]
    md.kind, md.mod_id, md.head_index = kind, mod_id, head_index
    md.pos = emb.pos_emb.data_ptr()
    md.mod_emb = emb.mod_emb.data_ptr()
    md.shifted = 1 if (is_dec and kind == L.KIND_SEQ and raw != 2) else 0
    md.max_len = getattr(emb, "max_length", 0) or 0
    if kind in (L.KIND_TOK, L.KIND_SEQ):
        ids = t.reshape(B, -1)
        if ids.dtype not in (torch.int32, torch.int64):
            ids = ids.long()
        ids = ids.contiguous()
        keep.append(ids)
        md.ids, md.ids_are_i64, md.id_stride = ids.data_ptr(), 1 if ids.dtype == torch.int64 else 0, ids.shape[1]
        md.table = emb.token_emb.weight.data_ptr()
        md.L = ids.shape[1] - (1 if md.shifted else 0)
        if ids.shape[1] != mask.shape[1]:
            raise ValueError(f"{name}: tensor has {ids.shape[1]} positions but the mask has {mask.shape[1]}")
    elif kind == L.KIND_PATCH:
        px = t.float().contiguous()
        keep.append(px)
        _, C, Hh, Ww = px.shape
        ps = emb.patch_size[0]
        md.ids, md.id_stride = px.data_ptr(), C * Hh * Ww
        md.patch, md.channels, md.grid_w = ps, C, Ww // ps
        md.L = (Hh // ps) * (Ww // ps)
    else:  # KIND_SEQ_EMB
        e = t.float().contiguous()
        keep.append(e)
        md.ids, md.id_stride, md.orig_dim = e.data_ptr(), e.shape[1] * e.shape[2], e.shape[2]
        md.proj_bias = emb.emb_proj.bias.data_ptr()
        md.L = e.shape[1]
    if is_dec and (raw != 2 or "decoder_attention_mask" in d):
        dam = d["decoder_attention_mask"].reshape(B, -1)
        if dam.dtype != torch.int32:
            dam = dam.int()
        dam = dam.contiguous()
        keep.append(dam)
        md.dam = dam.data_ptr()
    return keep


class FourMEngine:
    _pending = None            # (engines built without this __init__ - the tokenizer's - start with no deferred residual)
    _drop_now = None           # (per-sample DropPath scale, rows per sample) of the residual branch being computed
    drop_uniforms = None       # tests: an iterator of (B,) uniform tensors replacing torch.rand in DropPath

    def __init__(self, model):
        from fourm.models.fm_utils import GatedMlp, NormAttention, act_name
        self.model = model
        self.D = model.dim
        blk = model.encoder[0] if len(model.encoder) else model.decoder[0]
        attn0 = blk.attn if hasattr(blk, "attn") else blk.self_attn
        self.H = attn0.num_heads
        if self.D // self.H != 64:
            raise NotImplementedError(f"head_dim {self.D // self.H}: the HIP attention kernels are built for head_dim 64")
        self.qk_norm = isinstance(attn0, NormAttention)       # per-head LayerNorm on q / k (fm_utils.py:222-308)
        self.gated = isinstance(blk.mlp, GatedMlp)
        self.act = act_name(blk.mlp.act)
        if self.gated and self.act != "silu":
            raise NotImplementedError("gated MLP is implemented for SiLU (SwiGLU) only")
        if not self.gated and self.act != "gelu":
            raise NotImplementedError("plain MLP is implemented for exact GELU only")
        self.Hd = blk.mlp.hidden_features
        self.Hp = ru(self.Hd, 64)
        self.scale = 64 ** -0.5
        self.eps = blk.norm1.eps
        # "bf16" = the hot path (autocast semantics); "fp32" = the verification path (csrc/fp32_verify.hip): fp32 activations and
        # weights through plain kernels, no rounding anywhere - same launch sequence, checked against the upstream fp32 model
        prec = getattr(model, "compute_precision", None) or os.environ.get("FOURM_PRECISION", "bf16")
        if prec not in ("bf16", "fp32"):
            raise ValueError(f"compute_precision {prec!r}: 'bf16' or 'fp32'")
        self.fp32 = prec == "fp32"
        self.adt = torch.float32 if self.fp32 else torch.bfloat16
        self.ws: Optional[Workspace] = None
        self.shadows: Dict[tuple, Shadow] = {}
        self._shadow_table = None
        self.flat_params = self.flat_grads = None
        self._slices = {}
        self._ctx = None           # saved state of the last training forward
        self._dw_jobs = None       # weight-gradient GEMMs queued by the running block backward (None: launch each at once)
        self._pending = None       # (stream buffer still to be written, residual input, bf16 delta): see _residual / _ln
        self.reducer = None        # fourm.parallel.GradReducer when gradients are exchanged (data parallel)

    @property
    def checkpointing(self):
        return bool(getattr(self.model, "use_act_checkpoint", False))

    # ------------------------------------------------------------------------------------------
    # flat parameter / gradient stores
    # ------------------------------------------------------------------------------------------
    @property
    def device(self):
        return self.model.mask_token.device

    def _unique_params(self):
        seen, out = set(), []
        for n, p in self.model.named_parameters():
            if id(p) not in seen:
                seen.add(id(p))
                out.append((n, p))
        return out

    def flatten(self):
        """Move every parameter into one flat fp32 buffer (decay-type tensors first, then norm/bias
        tensors, each 64-byte aligned) and create the gradient store with the same layout."""
        named = self._unique_params()
        dev = self.device

        def nodecay(n):
            return "norm." in n or ".norm" in n or n.endswith(".bias")
        # (context-norm hoist: the decoder blocks' cross_attn.kv weights get their gradient in one piece after the decoder loop - they
        # sit together behind the other decay-type tensors so that their slice of the gradient store is ONE range of one stage)
        late = {id(b.cross_attn.kv.weight) for b in self.model.decoder} if self.hoist_ctx else set()
        named.sort(key=lambda np_: (nodecay(np_[0]), id(np_[1]) in late))
        off, slices = 0, {}
        for n, p in named:
            slices[id(p)] = (off, p.numel())
            off += ru(p.numel(), 16)
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        for n, p in named:
            o, k = slices[id(p)]
            flat[o:o + k].copy_(p.data.reshape(-1))
            p.data = flat[o:o + k].view(p.shape)
        self.flat_params, self._slices = flat, slices
        self.flat_grads = torch.zeros_like(flat)
        self._named = named
        bump_weight_epoch()

    def _ensure_flat(self):
        if self.flat_params is None or self.flat_params.device != self.device:
            self.flatten()
            return
        for n, p in self._named[:4] + self._named[-4:]:      # cheap check that nobody re-pointed .data
            o, k = self._slices[id(p)]
            if p.data_ptr() != self.flat_params.data_ptr() + 4 * o:
                self.flatten()
                return

    def grad_stages(self):
        """{stage: [(offset, length)]} over the flat gradient store, keyed by the backward stage after
        which those gradients are final.  Shared tensors (tied heads, shared mod_emb, norm weights in the
        trailing no-decay region) belong to the last stage that touches them."""
        m = self.model
        owner = {}

        def claim(stage, module_or_params):
            ps = module_or_params.parameters() if isinstance(module_or_params, nn.Module) else module_or_params
            for p in ps:
                owner[id(p)] = stage          # later claims win
        for i, blk in enumerate(m.decoder):
            claim(f"dec{i}", blk)
        if self.hoist_ctx:          # final only when the folded gradients have been unfolded (train_backward, after the decoder loop)
            claim("dec_emb", [b.cross_attn.kv.weight for b in m.decoder] + [b.context_norm.weight for b in m.decoder])
        claim("heads", [m.decoder_norm.weight] + ([m.decoder_norm.bias] if isinstance(m.decoder_norm.bias, nn.Parameter) else []))
        claim("dec_emb", m.decoder_embeddings)
        claim("dec_emb", [m.mask_token])
        claim("ctx", m.decoder_proj_context)
        claim("ctx", m.encoder_norm)
        for i, blk in enumerate(m.encoder):
            claim(f"enc{i}", blk)
        claim("enc_emb", m.encoder_embeddings)       # includes the mod_emb shared with the decoder side
        if m.register_tokens is not None:
            claim("enc_emb", [m.register_tokens])
        stages = {}
        for n, p in self._named:
            o, k = self._slices[id(p)]
            # tiny tensors (norm weights, biases) all sit in the trailing no-decay region: one final slice
            stage = owner.get(id(p), "enc_emb") if k >= 65536 else "tail"
            stages.setdefault(stage, []).append((o, ru(k, 16)))
        return stages

    def _stage(self, name):
        if self.reducer is not None:
            self.reducer.stage_done(name)

    def grad_view(self, p):
        o, k = self._slices[id(p)]
        return self.flat_grads[o:o + k].view(p.shape)

    def grads_were_cleared(self) -> bool:
        """True when the parameters that carried a gradient view after the previous backward no longer do (the trainer ran
        ``zero_grad(set_to_none=True)``, or this is the first backward): the flat store must be zeroed before accumulating."""
        live = getattr(self, "_attached", None)
        if not live:
            return True
        for p in live[:4] + live[-4:]:
            if p.grad is None or p.grad.data_ptr() != self.grad_view(p).data_ptr():
                return True
        return False

    def untouched_params(self):
        """Parameters no token of the last training forward can reach: embeddings / heads of modalities absent from
        ``mod_dict``.  Upstream leaves their ``.grad`` None (autograd never sees them), so AdamW skips them - no weight
        decay, no moment decay, no step count.  With a gradient reducer attached every rank must take the same optimizer
        step, so nothing is skipped there (DDP itself only skips parameters unused on EVERY rank)."""
        c = self._ctx
        if c is None or self.reducer is not None:
            return []
        m, touched, cand = self.model, set(), []
        for names, embs in ((c["enc"]["names"], m.encoder_embeddings), (c["dec"]["names"], m.decoder_embeddings)):
            for n, e in embs.items():
                if n in names:
                    touched.update(id(p) for p in e.parameters())
                else:
                    cand.extend(e.parameters())
        return [p for p in cand if id(p) not in touched]

    def attach_grads(self, zero: bool, untouched=()):
        """param.grad := view into the flat gradient store.  ``zero`` clears the store first (a fresh
        accumulation window: the trainer called optimizer.zero_grad()).  ``untouched`` parameters (see
        ``_untouched_params``) keep ``grad = None`` unless an earlier micro-batch of the window reached them."""
        if zero:
            with ops._prof("fills"):
                self.flat_grads.zero_()
            self._window_touched = set()
        skip = {id(p) for p in untouched} - getattr(self, "_window_touched", set())
        self._attached = []
        for n, p in self._named:
            if not p.requires_grad:
                continue
            if id(p) in skip:
                p.grad = None
                continue
            self._window_touched.add(id(p))
            p.grad = self.grad_view(p)
            self._attached.append(p)

    # ------------------------------------------------------------------------------------------
    # weight shadows
    # ------------------------------------------------------------------------------------------
    def _stamp(self, params):
        return tuple((p._version, _WEIGHT_EPOCH, p.data_ptr()) for p in params)

    def _get_shadow(self, key, make) -> torch.Tensor:
        s = self.shadows.get(key)
        if s is None:
            s = self.shadows[key] = make()
            for p in s.params:       # FusedAdamW rewrites these copies while it updates the master (fm_adamw_shadow)
                regs = getattr(p, "_fourm_shadows", None)
                if regs is None:
                    regs = p._fourm_shadows = []
                regs.append((weakref.ref(self), key))
        if s.stamp != self._stamp(s.params):
            self._refresh_shadows()
        return s.buf

    def _refresh_shadows(self):
        """Rebuild EVERY stale shadow in one launch (after an optimizer step that is all of them: one table
        walk instead of ~250 small cast / transpose launches)."""
        stale = [s for s in self.shadows.values() if s.stamp != self._stamp(s.params)]
        jobs = []
        for s in stale:
            jobs += [(j[0].detach().reshape(j[0].shape[0], -1), j[1], j[2], j[3].detach() if len(j) > 3 and j[3] is not None else None) for j in s.jobs]
        sig = tuple((a.data_ptr(), b.data_ptr(), tr, sc.data_ptr() if sc is not None else 0) for (a, b, tr, sc) in jobs)
        if self._shadow_table is None or self._shadow_table[0] != sig:
            table, tiles = ops.shadow_jobs_table(jobs, self.device)
            self._shadow_table = (sig, table, len(jobs), tiles)
        _, table, n, tiles = self._shadow_table
        ops.shadow_refresh(table, n, tiles)
        for s in stale:
            s.stamp = self._stamp(s.params)

    def mark_shadows_fresh(self, keys, written):
        """Called by FusedAdamW after a fused update: shadow ``key`` is current when every one of its (master, destination)
        jobs is in ``written`` = {(id(param), destination data_ptr)}."""
        for key in keys:
            s = self.shadows.get(key)
            if s is not None and all(len(j) == 3 and (id(j[0]), j[1].data_ptr()) in written for j in s.jobs):
                s.stamp = self._stamp(s.params)

    def w(self, p, pad_rows=False):
        """(out, in_p) bf16, pad columns zero: the W operand of y = x W^T.  pad_rows: the image has ru(out, 64) rows, the extra ones zero
        (4M-L's hidden width 2730 -> 2752: the GEMM then runs over whole 64-column groups and writes zeros into the pad columns)."""
        if self.fp32:
            return p.detach().reshape(p.shape[0], -1)
        def make():
            out_f, in_f = p.shape[0], p[0].numel()
            buf = torch.zeros(ru(out_f, 64) if pad_rows else out_f, ru(in_f, 64), dtype=torch.bfloat16, device=p.device)
            return Shadow(buf, (p,), [(p, buf, False)])
        s = self._get_shadow(("w", id(p)), make)
        # one image per weight: the first request fixes its row count; a padded request on an unpadded image would let the GEMM read past it
        assert not pad_rows or s.shape[0] >= ru(p.shape[0], 64), "w(): the cached image was created without pad_rows"
        return s

    def wt(self, p, pad_rows=False):
        """(in, out_p) bf16, pad columns zero: the W operand of dX = dY W.  pad_rows: ru(in, 64) rows, the extra ones zero."""
        if self.fp32:
            return p.detach().reshape(p.shape[0], -1).t()      # strided view: the fp32 GEMM takes any strides
        def make():
            out_f, in_f = p.shape[0], p[0].numel()
            buf = torch.zeros(ru(in_f, 64) if pad_rows else in_f, ru(out_f, 64), dtype=torch.bfloat16, device=p.device)
            return Shadow(buf, (p,), [(p, buf, True)])
        s = self._get_shadow(("wt", id(p)), make)
        assert not pad_rows or s.shape[0] >= ru(p[0].numel(), 64), "wt(): the cached image was created without pad_rows"
        return s

    def w_fold(self, lin, norm):
        """(out, in_p) image of lin.weight * norm.weight[None, :] - the W operand of y = x_hat (W diag(gamma))^T (context-norm hoist)."""
        p, g = lin.weight, norm.weight
        def make():
            out_f, in_f = p.shape
            buf = torch.zeros(out_f, in_f if self.fp32 else ru(in_f, 64), dtype=self.adt, device=p.device)
            return Shadow(buf, (p, g), [(p, buf, False, g)])
        return self._get_shadow(("wfold", id(p)), make)

    def wt_fold_stack(self):
        """(D, L * 2D) image [ (W_0 diag(gamma_0))^T | (W_1 diag(gamma_1))^T | ... ] of every decoder block's cross_attn.kv weight:
        the W operand of d(x_hat) = [dkv_0 | dkv_1 | ...] W_stack, the sum over the layers as ONE reduction of length L * 2D."""
        m, D = self.model, self.D
        def make():
            Ld = len(m.decoder)
            buf = torch.zeros(D, Ld * 2 * D, dtype=self.adt, device=self.device)
            ps, jobs = [], []
            for i, blk in enumerate(m.decoder):
                p, g = blk.cross_attn.kv.weight, blk.context_norm.weight
                ps += [p, g]
                jobs.append((p, buf[:, i * 2 * D:(i + 1) * 2 * D], True, g))
            return Shadow(buf, tuple(ps), jobs)
        return self._get_shadow(("wtfold", 0), make)

    @property
    def hoist_ctx(self):
        """The context-norm hoist applies: every decoder block's context_norm is bias-free and its K/V projection has no bias (all 4M
        swiglu_nobias configurations).  Decided once per engine (it fixes the order of the flat parameter store); which of the two tensors
        are trainable is read at every backward."""
        h = getattr(self, "_hoist_ctx", None)
        if h is None:
            m = self.model
            h = HOIST_CTX and len(m.decoder) > 0 and all(
                not isinstance(b.context_norm.bias, nn.Parameter) and b.cross_attn.kv.bias is None for b in m.decoder)
            self._hoist_ctx = h
        return h

    def w13t(self, mlp):
        """(D, 2*Hp) bf16 = [fc1^T | fc3^T]: the W operand of d(h2) = [dg | du] [fc1; fc3]."""
        def make():
            buf = torch.zeros(self.D, 2 * self.Hp, dtype=torch.bfloat16, device=self.device)
            p1, p3 = mlp.fc1.weight, mlp.fc3.weight
            return Shadow(buf, (p1, p3), [(p1, buf[:, :self.Hp], True), (p3, buf[:, self.Hp:], True)])
        return self._get_shadow(("w13t", id(mlp)), make)

    # ------------------------------------------------------------------------------------------
    # selection + embedding
    # ------------------------------------------------------------------------------------------
    def _mod_desc(self, md: L.ModDesc, name, d, emb, is_dec, head_index=0, raw=0):
        return fill_mod_desc(md, d, emb, is_dec, int(self.model.modality_info[name]["id"]), head_index, raw, name)

    def select(self, mod_dict, n_keep: int, is_dec: bool, order: List[str], prefix: str, want_x0=True, heads=None, raw=0):
        """Run the fused concat/partition/embed kernel for one side.  Returns a dict of device tensors.
        With ``want_x0`` the dense projections (pixels, T5 embeddings) are added to ``x0 = tokens + emb``
        (what the trunk consumes); without it they are added to ``tokens`` (the upstream sub-API view).
        ``raw`` = 1: every concatenated position in place, nothing zeroed (cat_*_tensors; ``n_keep`` is ignored)."""
        m = self.model
        embs = m.decoder_embeddings if is_dec else m.encoder_embeddings
        names = [n for n in order if n in embs]
        if not names:
            raise ValueError("no modality of mod_dict is known to the model")
        if len(names) > L.FM_MAX_MODS:
            raise ValueError(f"{len(names)} modalities exceed FM_MAX_MODS={L.FM_MAX_MODS}")
        B = mod_dict[names[0]]["tensor"].shape[0]
        D, ws = self.D, self.ws
        n_reg = 0 if (is_dec or raw) else m.num_register_tokens
        desc = L.SelectDesc()
        keep, total = [], 0
        # heads = decoder modalities present in this batch, in mod_dict order (fm.py:669-671, :590)
        head_names = heads if heads is not None else [n for n in mod_dict if n in m.decoder_embeddings]
        out_heads = head_names
        patch_ld = seq_ld = 0
        for i, n in enumerate(names):
            keep += self._mod_desc(desc.mods[i], n, mod_dict[n], embs[n], is_dec, head_names.index(n) if is_dec else 0, raw)
            total += desc.mods[i].L
            if desc.mods[i].kind == L.KIND_PATCH:
                patch_ld = max(patch_ld, ru(embs[n].proj.weight.shape[1], 64))
            if desc.mods[i].kind == L.KIND_SEQ_EMB:
                seq_ld = max(seq_ld, ru(embs[n].orig_emb_dim, 64))
        if raw:
            n_keep = total
        Nt = n_reg + n_keep
        R, Rp = B * Nt, ru(B * Nt, 128)
        desc.raw = raw
        desc.n_mods, desc.batch, desc.dim, desc.n_keep, desc.n_reg = len(names), B, D, n_keep, n_reg
        desc.total_len, desc.is_decoder = total, 1 if is_dec else 0
        if n_keep > total:
            raise ValueError(f"asked to keep {n_keep} tokens but the batch only has {total} positions")
        out = dict(B=B, Nt=Nt, R=R, names=names, heads=out_heads)
        f32, dev = torch.float32, self.device
        out["tokens"] = ws.get(prefix + "tokens", (Rp, D), f32)
        out["emb"] = ws.get(prefix + "emb", (Rp, D), f32)
        out["x0"] = ws.get(prefix + "x0", (Rp, D), f32) if want_x0 else None
        out["mask"] = ws.get(prefix + "mask", (B, Nt), torch.bool)
        out["mod_mask"] = ws.get(prefix + "mod_mask", (B, Nt), torch.int16)
        out["slot_mod"] = ws.get(prefix + "slot_mod", (B, Nt), torch.int32)
        out["slot_src"] = ws.get(prefix + "slot_src", (B, Nt), torch.int32)
        out["slot_pos"] = ws.get(prefix + "slot_pos", (B, Nt), torch.int32)
        desc.tokens, desc.emb = out["tokens"].data_ptr(), out["emb"].data_ptr()
        desc.x0 = out["x0"].data_ptr() if want_x0 else None
        desc.out_mask, desc.out_mod = out["mask"].data_ptr(), out["mod_mask"].data_ptr()
        desc.slot_mod, desc.slot_src, desc.slot_pos = out["slot_mod"].data_ptr(), out["slot_src"].data_ptr(), out["slot_pos"].data_ptr()
        if n_reg:
            desc.reg_tokens = m.register_tokens.data_ptr()
        if is_dec:
            out["target_ids"] = ws.get(prefix + "target_ids", (B, Nt), torch.int64)
            out["cs"] = ws.get(prefix + "cs", (B, Nt), torch.int32)
            out["mod_pre"] = ws.get(prefix + "mod_pre", (B, Nt), torch.int16)
            out["head_of_row"] = ws.get(prefix + "head_of_row", (B, Nt), torch.int32)
            desc.mask_token = m.mask_token.data_ptr()
            desc.target_ids, desc.out_cs = out["target_ids"].data_ptr(), out["cs"].data_ptr()
            desc.out_mod_pre, desc.out_mod_index = out["mod_pre"].data_ptr(), out["head_of_row"].data_ptr()
        if patch_ld:
            out["patch_rows"] = ws.get(prefix + "patch_rows", (Rp, patch_ld), self.adt)
            desc.patch_rows, desc.patch_ld = out["patch_rows"].data_ptr(), patch_ld
        if seq_ld:
            out["seqemb_rows"] = ws.get(prefix + "seqemb_rows", (Rp, seq_ld), self.adt)
            desc.seqemb_rows, desc.seqemb_ld = out["seqemb_rows"].data_ptr(), seq_ld
        desc.rows_f32 = 1 if self.fp32 else 0
        with ops._prof("select_embed"):
            L.check(L.select_embed(ops.C.byref(desc), ops._stream()))
        out["_keep"] = keep
        # dense projections of pixel / embedding modalities, added onto the (zero) token rows
        for n in names:
            e = embs[n]
            if e.kind == L.KIND_PATCH:
                dst = out["x0"] if want_x0 else out["tokens"]
                ops.gemm_nt(out["patch_rows"], self.w(e.proj.weight), dst, epilogue=L.EPI_RESIDUAL, res=dst, M=R, N=D)
            elif e.kind == L.KIND_SEQ_EMB:
                dst = out["x0"] if want_x0 else out["tokens"]
                ops.gemm_nt(out["seqemb_rows"], self.w(e.emb_proj.weight), dst, epilogue=L.EPI_RESIDUAL, res=dst, M=R, N=D)
        return out

    # ------------------------------------------------------------------------------------------
    # building blocks (forward).  `sv` is the per-layer dict that keeps what backward needs.
    # ------------------------------------------------------------------------------------------
    def _ln(self, norm, x, y, R, sv=None, key=None, tag="", row_map=None):
        mean = rstd = None
        if sv is not None:
            mean = self.ws.get(f"{tag}.{key}.mu", (x.shape[0],), torch.float32)
            rstd = self.ws.get(f"{tag}.{key}.rs", (x.shape[0],), torch.float32)
            sv[key + ".mu"], sv[key + ".rs"] = mean, rstd
            if LN_BWD_FROM_H and row_map is None and not isinstance(norm.bias, nn.Parameter) and y.dtype == torch.bfloat16:      # (a buffer bias is upstream's all-zero placeholder, fm_utils.py:93-108)
                sv[key + ".h"] = y          # the backward rebuilds x_hat from this 2-byte output instead of the 4-byte input (_ln_bwd)
        pend = self._pending
        if pend is not None and pend[0] is x:      # x = residual + delta is still owed: this norm computes and stores it on the way
            self._pending = None
            ops.layernorm_fwd(pend[1], norm.weight, norm.bias, y, mean, rstd, row_map=row_map, eps=norm.eps, R=R, delta=pend[2], x_out=x)
            return y
        self._settle()
        ops.layernorm_fwd(x, norm.weight, norm.bias, y, mean, rstd, row_map=row_map, eps=norm.eps, R=R)
        return y

    def _residual(self, a, lin, x_res, x_out, R, N, K, defer):
        """x_out = x_res + a W^T (+ bias): fused in the GEMM epilogue, or (defer) a bf16 GEMM whose sum is owed to the next _ln(x_out).
        With stochastic depth active (self._drop_now = (per-sample scale, rows per sample)) the branch output is scaled first."""
        drop = self._drop_now
        if drop is not None and self.fp32:
            raise NotImplementedError("drop_path in the fp32 verification mode")
        if drop is not None or (defer and DEFER_RESIDUAL and not self.fp32):
            self._settle()
            delta = self.ws.get("fwd.delta", (x_out.shape[0], N), self.adt)
            ops.gemm_nt(a, self.w(lin.weight), delta, bias=lin.bias, M=R, N=N, K=K)
            if drop is not None:
                ops.scale_rows_bf16(delta, drop[0], drop[1], R, N)
            self._pending = (x_out, x_res, delta, R)
            if not (defer and DEFER_RESIDUAL):
                self._settle()
        else:
            ops.gemm_nt(a, self.w(lin.weight), x_out, epilogue=L.EPI_RESIDUAL, res=x_res, bias=lin.bias, M=R, N=N, K=K)

    def _drop_scales(self, blk, B, n, sv, dp):
        """Per-branch DropPath scales of one block: ``dp`` when given (checkpoint recompute), freshly drawn in training mode, else None."""
        from fourm.models.fm_utils import DropPath
        if dp is None and isinstance(getattr(blk, "drop_path", None), DropPath) and self.model.training:
            dp = [blk.drop_path.sample_scale(B, self.device, None if self.drop_uniforms is None else next(self.drop_uniforms)) for _ in range(n)]
        if sv is not None and dp is not None:
            sv["dp"] = dp
        self._last_dp = dp
        return dp

    def _unit_norm(self):
        """LayerNorm without an affine part (weight 1, no bias) at the decoder blocks' context_norm epsilon: x_hat of the hoisted context norm."""
        n = getattr(self, "_unit", None)
        if n is None or n.weight.device != self.device:
            from types import SimpleNamespace
            n = self._unit = SimpleNamespace(weight=torch.ones(self.D, dtype=torch.float32, device=self.device), bias=None,
                                             eps=self.model.decoder[0].context_norm.eps)
        return n

    def _settle(self):
        """A deferred residual sum nobody normalised (a caller outside the trunk loops wants the stream itself): write it now."""
        pend, self._pending = self._pending, None
        if pend is not None:
            x_out, x_res, delta, R = pend
            ops.add_bf16_to_f32(x_res, delta, x_out, R)

    def _buf(self, sv, tag, key, shape, dtype):
        """Per-layer buffer when saving for backward, shared scratch otherwise."""
        name = f"{tag}.{key}" if sv is not None else f"scratch.{key}"
        t = self.ws.get(name, shape, dtype)
        if sv is not None:
            sv[key] = t
        return t

    def _mlp_fwd(self, mlp, h, x_res, x_out, R, Rp, sv, tag, defer=False):
        bf = self.adt
        if self.gated:
            gu = self._buf(sv, tag, "gu", (Rp, 2 * self.Hp), bf) if sv is not None else None       # inference: nothing to save
            act = self._buf(sv, tag, "act", (Rp, self.Hp), bf)
            # (bias-free, bf16: the weight images carry zero rows up to Hp, so a hidden width like 4M-L's 2730 runs as 2752 on the lock-step kernel;
            # the pad columns of act / gu receive silu(0) * 0 = 0, what they hold anyway)
            padN = not self.fp32 and mlp.fc1.bias is None and mlp.fc3.bias is None and self.Hp != self.Hd
            ops.gemm_nt(h, self.w(mlp.fc1.weight, padN), act, epilogue=L.EPI_SWIGLU, w2=self.w(mlp.fc3.weight, padN), out2=gu, Hp=self.Hp,
                        bias=mlp.fc1.bias, bias2=mlp.fc3.bias, M=R, N=self.Hp if padN else self.Hd, K=self.D)
        else:
            pre = self._buf(sv, tag, "pre", (Rp, self.Hp), bf) if sv is not None else None     # inference: nothing to save
            act = self._buf(sv, tag, "act", (Rp, self.Hp), bf)
            ops.gemm_nt(h, self.w(mlp.fc1.weight), act, epilogue=L.EPI_GELU, out2=pre, bias=mlp.fc1.bias, M=R, N=self.Hd, K=self.D)
        self._residual(act, mlp.fc2, x_res, x_out, R, self.D, self.Hp, defer)

    def _qk_norm_fwd(self, attn, q, k, Rq, Rk, Rqp, Rkp, sv, tag, key):
        """q_norm / k_norm of NormAttention / NormCrossAttention: bf16 q, k -> normalised bf16 copies + (mean, rstd)."""
        bf, f32, D, H = self.adt, torch.float32, self.D, self.H
        qn = self._buf(sv, tag, key + ".q", (Rqp, D), bf)
        kn = self._buf(sv, tag, key + ".k", (Rkp, D), bf)
        sq = self._buf(sv, tag, key + ".sq", (Rqp * H, 2), f32)
        sk = self._buf(sv, tag, key + ".sk", (Rkp * H, 2), f32)
        ops.headnorm_fwd(q, attn.q_norm.weight, attn.q_norm.bias, qn, sq, Rq, H, attn.q_norm.eps)
        ops.headnorm_fwd(k, attn.k_norm.weight, attn.k_norm.bias, kn, sk, Rk, H, attn.k_norm.eps)
        return qn, kn

    def _qk_norm_bwd(self, attn, sv, key, dqn, dkn, q, k, dq, dk, Rq, Rk):
        """dq / dk (bf16) from the gradients of the normalised copies; accumulates q_norm / k_norm weight (bias) grads."""
        for norm, d_n, x, d_x, st, R in ((attn.q_norm, dqn, q, dq, sv[key + ".sq"], Rq), (attn.k_norm, dkn, k, dk, sv[key + ".sk"], Rk)):
            db = self._g(norm.bias) if isinstance(norm.bias, nn.Parameter) else None
            ops.headnorm_bwd(d_n, x, norm.weight, st, d_x, self._g(norm.weight), db, R, self.H)

    def _self_attn_fwd(self, attn, h, x_res, x_out, B, N, R, Rp, mask, sv, tag):
        bf, D = self.adt, self.D
        qkv = self._buf(sv, tag, "qkv", (Rp, 3 * D), bf)
        o = self._buf(sv, tag, "o", (Rp, D), bf)
        ops.gemm_nt(h, self.w(attn.qkv.weight), qkv, bias=attn.qkv.bias, M=R, N=3 * D, K=D)
        q_in, k_in = qkv[:, :D], qkv[:, D:2 * D]
        if self.qk_norm:
            q_in, k_in = self._qk_norm_fwd(attn, q_in, k_in, R, R, Rp, Rp, sv, tag, "qkn")
        sm = sl = None
        if sv is not None:
            sm = self._buf(sv, tag, "sm", (B, self.H, N), torch.float32)
            sl = self._buf(sv, tag, "sl", (B, self.H, N), torch.float32)
        ops.attn_fwd(q_in, k_in, qkv[:, 2 * D:], o, B, self.H, N, N, self.scale, stat_m=sm, stat_l=sl, zero_attn=getattr(attn, "allow_zero_attn", False), **mask)
        self._residual(o, attn.proj, x_res, x_out, R, D, D, defer=True)       # (every caller normalises x_out next)

    def _cross_attn_fwd(self, attn, hq, hc, x_res, x_out, B, M, N, Rq, Rqp, Rc, Rcp, mask, sv, tag, w_kv=None):
        bf, D = self.adt, self.D
        q = self._buf(sv, tag, "q", (Rqp, D), bf)
        kv = self._buf(sv, tag, "kv", (Rcp, 2 * D), bf)
        o = self._buf(sv, tag, "o2", (Rqp, D), bf)
        ops.gemm_nt(hq, self.w(attn.q.weight), q, bias=attn.q.bias, M=Rq, N=D, K=D)
        ops.gemm_nt(hc, self.w(attn.kv.weight) if w_kv is None else w_kv, kv, bias=attn.kv.bias, M=Rc, N=2 * D, K=D)
        sm = sl = None
        if sv is not None:
            sm = self._buf(sv, tag, "sm2", (B, self.H, M), torch.float32)
            sl = self._buf(sv, tag, "sl2", (B, self.H, M), torch.float32)
        q_in, k_in = q, kv[:, :D]
        if self.qk_norm:
            q_in, k_in = self._qk_norm_fwd(attn, q_in, k_in, Rq, Rc, Rqp, Rcp, sv, tag, "xqkn")
        ops.attn_fwd(q_in, k_in, kv[:, D:], o, B, self.H, M, N, self.scale, stat_m=sm, stat_l=sl, zero_attn=getattr(attn, "allow_zero_attn", False), **mask)
        self._residual(o, attn.proj, x_res, x_out, Rq, D, D, defer=True)

    def encoder_block_fwd(self, blk, x_in, B, N, mask, sv, tag, defer_out=False, out_name=None, dp=None):
        """x_in (Rp, D) f32 -> new (Rp, D) f32 buffer.  [upstream Block.forward, fm_utils.py:331-334]
        defer_out (trunk loops only): the block's last residual sum is left to the LayerNorm that consumes the returned buffer next."""
        R, Rp, D = B * N, x_in.shape[0], self.D
        bf, f32 = self.adt, torch.float32
        dp = self._drop_scales(blk, B, 2, sv, dp)
        h1 = self._ln(blk.norm1, x_in, self._buf(sv, tag, "h1", (Rp, D), bf), R, sv, "n1", tag)
        x_mid = self._buf(sv, tag, "x_mid", (Rp, D), f32)
        self._drop_now = (dp[0], N) if dp else None
        self._self_attn_fwd(blk.attn, h1, x_in, x_mid, B, N, R, Rp, mask, sv, tag)
        h2 = self._ln(blk.norm2, x_mid, self._buf(sv, tag, "h2", (Rp, D), bf), R, sv, "n2", tag)
        x_out = self.ws.get(out_name or (tag + ".x_out" if sv is not None else "scratch.x_out" + tag[-1:]), (Rp, D), f32)
        self._drop_now = (dp[1], N) if dp else None
        self._mlp_fwd(blk.mlp, h2, x_mid, x_out, R, Rp, sv, tag, defer=defer_out)
        self._drop_now = None
        if sv is not None:
            sv["x_in"] = x_in
        return x_out

    def decoder_block_fwd(self, blk, y_in, ctx, B, M, N, sa_mask, xa_mask, sv, tag, defer_out=False, out_name=None, dp=None, ctx_hat=None):
        """[upstream DecoderBlock.forward, fm_utils.py:362-366]
        ctx_hat (trunk loops with the context-norm hoist): the normalised context WITHOUT the affine part, computed once for all blocks;
        this block's context_norm.weight is then inside the kv weight image (w_fold)."""
        Rq, Rqp, Rc, Rcp, D = B * M, y_in.shape[0], B * N, ctx.shape[0], self.D
        bf, f32 = self.adt, torch.float32
        dp = self._drop_scales(blk, B, 3, sv, dp)
        h1 = self._ln(blk.norm1, y_in, self._buf(sv, tag, "h1", (Rqp, D), bf), Rq, sv, "n1", tag)
        y1 = self._buf(sv, tag, "y1", (Rqp, D), f32)
        self._drop_now = (dp[0], M) if dp else None
        self._self_attn_fwd(blk.self_attn, h1, y_in, y1, B, M, Rq, Rqp, sa_mask, sv, tag)
        self._drop_now = (dp[1], M) if dp else None
        hq = self._ln(blk.query_norm, y1, self._buf(sv, tag, "hq", (Rqp, D), bf), Rq, sv, "nq", tag)
        if ctx_hat is None:
            hc, w_kv = self._ln(blk.context_norm, ctx, self._buf(sv, tag, "hc", (Rcp, D), bf), Rc, sv, "nc", tag), None
        else:
            hc, w_kv = ctx_hat, self.w_fold(blk.cross_attn.kv, blk.context_norm)
        y2 = self._buf(sv, tag, "y2", (Rqp, D), f32)
        self._cross_attn_fwd(blk.cross_attn, hq, hc, y1, y2, B, M, N, Rq, Rqp, Rc, Rcp, xa_mask, sv, tag, w_kv=w_kv)
        h2 = self._ln(blk.norm2, y2, self._buf(sv, tag, "h2", (Rqp, D), bf), Rq, sv, "n2", tag)
        y_out = self.ws.get(out_name or (tag + ".y_out" if sv is not None else "scratch.y_out" + tag[-1:]), (Rqp, D), f32)
        self._drop_now = (dp[2], M) if dp else None
        self._mlp_fwd(blk.mlp, h2, y2, y_out, Rq, Rqp, sv, tag, defer=defer_out)
        self._drop_now = None
        if sv is not None:
            sv["y_in"] = y_in
        return y_out

    # ------------------------------------------------------------------------------------------
    # whole-model forward
    # ------------------------------------------------------------------------------------------
    def prepare(self):
        if self.ws is None or self.ws.device != self.device:
            self.ws = Workspace(self.device)

    @staticmethod
    def keypad(mask_u8):
        return dict(mask_kind=L.MASK_KEYPAD, kpad=mask_u8)

    def decoder_mask(self, cs, mod_pre):
        m = self.model
        return dict(mask_kind=L.MASK_DECODER, cs=None if m.decoder_causal_mask else cs, causal=m.decoder_causal_mask,
                    modq=mod_pre if m.decoder_sep_mask else None, modk=mod_pre if m.decoder_sep_mask else None)

    def encode_context(self, enc, st=None):
        """Encoder + encoder_norm + context projection on the selected input tokens (fm.py:477-480, :679).
        -> (final encoder stream f32, context f32 (Rp, D), key-padding mask kwargs, saved top-level state | None)."""
        m = self.model
        save = st is not None
        B, N = enc["B"], enc["Nt"]
        emask = self.keypad(enc["mask"])
        x = enc["x0"]
        ckpt = save and self.checkpointing
        for i, blk in enumerate(m.encoder):
            if ckpt:      # activation checkpointing (fm.py:103-113, use_act_checkpoint): keep the block's INPUT only, recompute in the backward
                st["enc_layers"].append(dict(ckpt_in=x))
                x = self.encoder_block_fwd(blk, x, B, N, emask, None, f"enc{i % 2}", defer_out=True, out_name=f"enc{i}.x_out")
                st["enc_layers"][-1]["dp"] = self._last_dp
                continue
            sv = {} if save else None
            x = self.encoder_block_fwd(blk, x, B, N, emask, sv, f"enc{i}" if save else f"enc{i % 2}", defer_out=True)   # next: norm1 / encoder_norm
            if save:
                st["enc_layers"].append(sv)
        R, Rp, D = B * N, x.shape[0], self.D
        sv_top = {} if save else None
        xn = self._ln(m.encoder_norm, x, self._buf(sv_top, "top", "xn", (Rp, D), self.adt), R, sv_top, "en", "top")
        ctx = self._buf(sv_top, "top", "ctx", (Rp, D), torch.float32)
        pc = m.decoder_proj_context
        ops.gemm_nt(xn, self.w(pc.weight), ctx, epilogue=L.EPI_RESIDUAL, res=enc["emb"], bias=pc.bias, M=R, N=D, K=D)
        return x, ctx, emask, sv_top

    def trunk_forward(self, enc, dec, save: bool):
        """Encoder + context projection + decoder on the selected tokens.  Returns the final decoder
        residual stream (f32) and, when saving, the per-layer state."""
        m = self.model
        B, N, Mt = enc["B"], enc["Nt"], dec["Nt"]
        st = dict(enc_layers=[], dec_layers=[]) if save else None
        x, ctx, emask, sv_top = self.encode_context(enc, st)
        y = dec["x0"]
        smask = self.decoder_mask(dec["cs"], dec["mod_pre"]) if "cs" in dec else dec["sa_mask"]
        ctx_hat, sv_hat = None, None
        if self.hoist_ctx:        # x_hat of the context once for all decoder blocks (see HOIST_CTX)
            sv_hat = {} if save else None
            ctx_hat = self._ln(self._unit_norm(), ctx, self._buf(sv_hat, "top", "ctx_hat", (ctx.shape[0], self.D), self.adt), B * N, sv_hat, "ch", "top")
        for i, blk in enumerate(m.decoder):
            if save and self.checkpointing:
                st["dec_layers"].append(dict(ckpt_in=y))
                y = self.decoder_block_fwd(blk, y, ctx, B, Mt, N, smask, emask, None, f"dec{i % 2}", defer_out=i + 1 < len(m.decoder),
                                           out_name=f"dec{i}.y_out", ctx_hat=ctx_hat)
                st["dec_layers"][-1]["dp"] = self._last_dp
                continue
            sv = {} if save else None
            y = self.decoder_block_fwd(blk, y, ctx, B, Mt, N, smask, emask, sv, f"dec{i}" if save else f"dec{i % 2}",
                                       defer_out=i + 1 < len(m.decoder), ctx_hat=ctx_hat)           # next: the following block's norm1
            if save:
                st["dec_layers"].append(sv)
        if save:
            st.update(top=sv_top, x_final=x, y_final=y, ctx=ctx, emask=emask, smask=smask, ctx_hat=ctx_hat, sv_hat=sv_hat)
        return y, st

    def heads_setup(self, dec, y_final, save):
        """decoder_norm + bucketing of the decoder rows by modality head + grouped logits GEMM."""
        m = self.model
        heads = dec["heads"]
        nH = len(heads)
        B, Mt, D = dec["B"], dec["Nt"], self.D
        R = B * Mt
        Rp = ops.padded_rows(R, nH)
        ws, dev, i32 = self.ws, self.device, torch.int32
        hs = dict(n=nH, Rp=Rp, heads=heads)
        hs["seg_start"], hs["seg_count"] = ws.get("heads.seg_start", (nH,), i32), ws.get("heads.seg_count", (nH,), i32)
        hs["perm"], hs["r2p"] = ws.get("heads.perm", (Rp,), i32), ws.get("heads.r2p", (R,), i32)
        hs["tile_group"] = ws.get("heads.tile_group", (Rp // ops.SEG,), i32)
        ops.segment_rows(dec["head_of_row"].view(-1), nH, hs["seg_start"], hs["seg_count"], hs["perm"], hs["r2p"], hs["tile_group"])
        sv = {} if save else None
        yp = ws.get("heads.yp", (Rp, D), self.adt)
        # decoder_norm writes straight into the segmented layout; pad rows are cleared first
        with ops._prof("fills"):
            yp.zero_()
        self._ln(m.decoder_norm, y_final, yp, R, sv, "dn", "heads", row_map=hs["r2p"])
        vocabs = [m.decoder_embeddings[h].vocab_size for h in heads]
        hs["vocabs"], hs["maxV"] = vocabs, max(vocabs)
        ldl = ru(hs["maxV"], 64)
        w_fwd = [self.w(m.decoder_embeddings[h].to_logits.weight) for h in heads]       # refreshed if stale
        w_bwd = [self.wt(m.decoder_embeddings[h].to_logits.weight) for h in heads] if save else None
        key = tuple(t.data_ptr() for t in w_fwd) + (tuple(t.data_ptr() for t in w_bwd) if save else ())
        cache = getattr(self, "_head_groups", None)
        if cache is None or cache[0] != key:       # device tables of raw pointers: rebuilt only when a buffer moved
            fwd = ops.make_groups([dict(W=t, N=v, K=D, ldw=D) for t, v in zip(w_fwd, vocabs)], dev)
            if self.fp32:     # dY = d(logits) W reads the (V, D) master through strides: n = d (stride 1), k = v (stride D)
                bwd = ops.make_groups([dict(W=m.decoder_embeddings[h].to_logits.weight, N=D, K=v, ldw=D, transposed=1)
                                       for h, v in zip(heads, vocabs)], dev) if save else None
            else:
                bwd = ops.make_groups([dict(W=t, N=D, K=ru(v, 64), ldw=ru(v, 64)) for t, v in zip(w_bwd, vocabs)], dev) if save else None
            vt = torch.tensor(vocabs, dtype=i32, device=dev)
            self._head_groups = cache = (key, fwd, bwd, vt)
        hs["g_fwd"], hs["g_bwd"], hs["vocab_t"] = cache[1], cache[2], cache[3]
        logits = ws.get("heads.logits", (Rp, ldl), self.adt)
        if HEADS_DENSE and not self.fp32 and nH <= HEADS_DENSE_MAX and ops.heads_dense_ok(w_fwd, vocabs, logits):
            # one dense launch per head on gemm_nt3, row ranges read from seg_start / seg_count on the device (round 5: 1.03 -> ~0.8 ms at 4M-B)
            ops.gemm_nt_heads(yp, w_fwd, vocabs, hs["seg_start"], hs["seg_count"], logits, D)
        else:
            ops.gemm_nt_grouped(yp, hs["g_fwd"], hs["tile_group"], logits, hs["maxV"], max_K=D)
        hs.update(yp=yp, logits=logits, sv=sv)
        hs["row_loss"] = ws.get("heads.row_loss", (Rp,), torch.float32)
        hs["row_lse"] = ws.get("heads.row_lse", (Rp,), torch.float32)
        hs["head_loss"] = ws.get("heads.head_loss", (nH,), torch.float32)
        hs["total"] = ws.get("heads.total", (1,), torch.float32)
        return hs

    def loss_forward(self, dec, hs, loss_type):
        lt = L.LOSS_MOD if loss_type in ("mod", "modality") else L.LOSS_TOKEN
        hs["loss_type"] = lt
        ops.cross_entropy(hs["logits"], hs["perm"], hs["tile_group"], dec["target_ids"].view(-1), hs["vocab_t"], hs["seg_start"],
                          hs["seg_count"], hs["n"], hs["maxV"], hs["row_loss"], hs["row_lse"], hs["head_loss"], hs["total"], loss_type=lt)
        return hs["total"], hs["head_loss"]

    def dec_order(self, mod_dict):
        """Upstream shuffles the decoder modalities with random.sample at every forward (fm.py:306);
        the same call on the same RNG state yields the same order here."""
        names = [n for n in mod_dict if n in self.model.decoder_embeddings]
        return random.sample(names, len(names))

    def train_forward(self, mod_dict, n_enc, n_dec, loss_type, save=True):
        if loss_type not in ("mod", "modality", "token"):
            raise ValueError("Invalid loss type")
        self.prepare()
        if save:
            self._ensure_flat()
        enc_names = [n for n in mod_dict if n in self.model.encoder_embeddings]
        enc = self.select(mod_dict, n_enc, False, enc_names, "enc.")
        dec = self.select(mod_dict, n_dec, True, self.dec_order(mod_dict), "dec.")
        y, st = self.trunk_forward(enc, dec, save)
        hs = self.heads_setup(dec, y, save)
        total, head_loss = self.loss_forward(dec, hs, loss_type)
        if save:
            self._ctx = dict(enc=enc, dec=dec, st=st, hs=hs)
        return total, head_loss, hs["heads"], hs["seg_count"]

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def _g(self, p):
        """fp32 gradient accumulator of a parameter (view of the flat store), or None if frozen."""
        return self.grad_view(p) if p.requires_grad else None

    def _dW(self, dy, x, lin, R64, n_cols=None, dy_cols=None):
        """lin.weight.grad += dy^T x ; lin.bias.grad += colsum(dy)."""
        g = self._g(lin.weight)
        N = lin.weight.shape[0] if n_cols is None else n_cols
        if g is not None:
            if self._dw_jobs is not None:      # inside a block backward: one launch for the layer (_flush_dW)
                self._dw_jobs.append((dy, x, g.view(g.shape[0], -1), N, lin.weight[0].numel(), R64))
            else:
                ops.gemm_tn(dy, x, g.view(g.shape[0], -1), N=N, K=lin.weight[0].numel(), R=R64)
        if lin.bias is not None and lin.bias.requires_grad:
            ops.colsum(dy, self.grad_view(lin.bias), N, R=R64)

    def _flush_dW(self):
        """All weight gradients queued by the current block backward in ONE launch (fm_gemm_tn_multi).  Their operands - the
        bf16 output gradients and the saved activations of this layer - must still hold what they held when queued: the block
        backwards write each residual-stream gradient copy to its own buffer until this point."""
        jobs, self._dw_jobs = self._dw_jobs, None
        ops.gemm_tn_multi(jobs)

    def _ln_bwd(self, norm, dy, x, sv, key, g, g_bf, R, dres, dy_row_map=None):
        dw = self._g(norm.weight)
        db = self._g(norm.bias) if isinstance(norm.bias, nn.Parameter) else None
        ops.layernorm_bwd(dy, x, norm.weight, sv[key + ".mu"], sv[key + ".rs"], g, dres=dres, dx_bf16=g_bf, dw=dw, db=db,
                          dy_row_map=dy_row_map, R=R, h=sv.get(key + ".h"))

    def _mlp_bwd(self, mlp, sv, g_bf, R, Rp):
        """In: g_bf = d(out) bf16.  Out: dh (bf16 scratch) = gradient w.r.t. the norm2 output."""
        bf, D, Hp, Hd = self.adt, self.D, self.Hp, self.Hd
        R64 = R           # the TN kernel masks the reduction past the live rows itself (no reliance on zeroed padding)
        ws = self.ws
        self._dW(g_bf, sv["act"], mlp.fc2, R64)
        dh = ws.get("bwd.dh", (Rp, D), bf)
        # FUSE_ACT_BWD: the activation backward runs in the fc2 dX GEMM epilogue on the saved (g | u) / pre, so
        # d(act) never reaches HBM.  Off by default: even with 16-byte epilogue loads / stores the fused GEMM costs
        # 6.3 ms per 4M-B step against 3.1 + 3.0 ms for the plain GEMM + the coalesced stand-alone kernel.
        fuse = FUSE_ACT_BWD
        if not fuse:
            da = ws.get("bwd.da", (Rp, Hp), bf)
            padN = not self.fp32 and Hp != Hd                   # zero rows up to Hp in the weight image: d(act) of the pad columns is 0
            ops.gemm_nt(g_bf, self.wt(mlp.fc2.weight, padN), da, M=R, N=Hp if padN else Hd, K=D)
        if self.gated:
            dgu = ws.get("bwd.dgu", (Rp, 2 * Hp), bf)
            if fuse:
                ops.gemm_nt(g_bf, self.wt(mlp.fc2.weight), dgu, M=R, N=Hd, K=D, epilogue=L.EPI_SWIGLU_BWD, res=sv["gu"], Hp=Hp)
            else:
                ops.swiglu_bwd(da, sv["gu"], dgu, Hd, Hp, R=R)
            self._dW(dgu[:, :Hp], sv["h2"], mlp.fc1, R64, n_cols=Hd)
            self._dW(dgu[:, Hp:], sv["h2"], mlp.fc3, R64, n_cols=Hd)
            if self.fp32:     # d(h2) = dg fc1 + du fc3, straight from the two masters
                ops.gemm_nt(dgu[:, :Hp], self.wt(mlp.fc1.weight), dh, M=R, N=D, K=Hd)
                ops.gemm_nt(dgu[:, Hp:], self.wt(mlp.fc3.weight), dh, M=R, N=D, K=Hd, epilogue=L.EPI_F32, res=dh)
            else:
                ops.gemm_nt(dgu, self.w13t(mlp), dh, M=R, N=D, K=2 * Hp)
        else:
            dpre = ws.get("bwd.dpre", (Rp, Hp), bf)
            if fuse:
                ops.gemm_nt(g_bf, self.wt(mlp.fc2.weight), dpre, M=R, N=Hd, K=D, epilogue=L.EPI_GELU_BWD, res=sv["pre"])
            else:
                ops.gelu_bwd(da, sv["pre"], dpre, Hd, Hp, R=R)
            self._dW(dpre, sv["h2"], mlp.fc1, R64, n_cols=Hd)
            ops.gemm_nt(dpre, self.wt(mlp.fc1.weight), dh, M=R, N=D, K=Hp)
        return dh

    def _self_attn_bwd(self, attn, sv, g_bf, B, N, R, Rp, mask):
        bf, D = self.adt, self.D
        R64, ws = R, self.ws
        self._dW(g_bf, sv["o"], attn.proj, R64)
        do = ws.get("bwd.do", (Rp, D), bf)
        ops.gemm_nt(g_bf, self.wt(attn.proj.weight), do, M=R, N=D, K=D)
        dqkv = ws.get("bwd.dqkv", (Rp, 3 * D), bf)
        qkv = sv["qkv"]
        if self.qk_norm:
            dqkn = ws.get("bwd.dqkn", (Rp, 2 * D), bf)
            ops.attn_bwd(sv["qkn.q"], sv["qkn.k"], qkv[:, 2 * D:], sv["o"], do, dqkn[:, :D], dqkn[:, D:], dqkv[:, 2 * D:],
                         B, self.H, N, N, self.scale, sv["sm"], sv["sl"], zero_attn=getattr(attn, "allow_zero_attn", False), **mask)
            self._qk_norm_bwd(attn, sv, "qkn", dqkn[:, :D], dqkn[:, D:], qkv[:, :D], qkv[:, D:2 * D], dqkv[:, :D], dqkv[:, D:2 * D], R, R)
        else:
            ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], sv["o"], do, dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:],
                         B, self.H, N, N, self.scale, sv["sm"], sv["sl"], zero_attn=getattr(attn, "allow_zero_attn", False), **mask)
        self._dW(dqkv, sv["h1"], attn.qkv, R64)
        dh = ws.get("bwd.dh", (Rp, D), bf)
        ops.gemm_nt(dqkv, self.wt(attn.qkv.weight), dh, M=R, N=D, K=3 * D)
        return dh

    def encoder_block_bwd(self, blk, sv, g, g_bf, B, N, mask):
        R, Rp = B * N, g.shape[0]
        self._dw_jobs = []
        dp = sv.get("dp")                                                       # DropPath: the gradient entering a branch carries its scale
        if dp:
            ops.scale_rows_bf16(g_bf, dp[1], N, R)
        dh = self._mlp_bwd(blk.mlp, sv, g_bf, R, Rp)
        g1 = self.ws.get("bwd.enc.gbf1", tuple(g_bf.shape), g_bf.dtype)      # g_bf is still an operand of the queued fc2 dW
        self._ln_bwd(blk.norm2, dh, sv["x_mid"], sv, "n2", g, g1, R, dres=g)
        if dp:
            ops.scale_rows_bf16(g1, dp[0], N, R)
        dh = self._self_attn_bwd(blk.attn, sv, g1, B, N, R, Rp, mask)
        self._flush_dW()
        self._ln_bwd(blk.norm1, dh, sv["x_in"], sv, "n1", g, g_bf, R, dres=g)

    def decoder_block_bwd(self, blk, sv, g, g_bf, dctx, dctx_bf, ctx, B, M, N, sa_mask, xa_mask, hoist=None):
        """hoist (context-norm hoist) = (dkv of this layer as a column block of the all-layer buffer, x_hat, fp32 scratch for dL/dW'):
        the block then leaves d(kv) and dL/d(W diag(gamma)) behind; d(context) and the unfolding follow the decoder loop."""
        bf, D = self.adt, self.D
        Rq, Rqp, Rc, Rcp = B * M, g.shape[0], B * N, ctx.shape[0]
        ws = self.ws
        self._dw_jobs = []
        g_in = g_bf
        dp = sv.get("dp")
        if dp:
            ops.scale_rows_bf16(g_in, dp[2], M, Rq)
        dh = self._mlp_bwd(blk.mlp, sv, g_in, Rq, Rqp)
        g_bf = ws.get("bwd.dec.gbf1", tuple(g_in.shape), g_in.dtype)          # each copy stays an operand of a queued dW
        self._ln_bwd(blk.norm2, dh, sv["y2"], sv, "n2", g, g_bf, Rq, dres=g)
        if dp:
            ops.scale_rows_bf16(g_bf, dp[1], M, Rq)
        # cross attention
        xa = blk.cross_attn
        self._dW(g_bf, sv["o2"], xa.proj, Rq)
        do = ws.get("bwd.do", (Rqp, D), bf)
        ops.gemm_nt(g_bf, self.wt(xa.proj.weight), do, M=Rq, N=D, K=D)
        dq = ws.get("bwd.dq", (Rqp, D), bf)
        dkv = ws.get("bwd.dkv", (Rcp, 2 * D), bf) if hoist is None else hoist[0]
        kv = sv["kv"]
        if self.qk_norm:
            dqn = ws.get("bwd.dqn", (Rqp, D), bf)
            dkn = ws.get("bwd.dkn", (Rcp, D), bf)
            ops.attn_bwd(sv["xqkn.q"], sv["xqkn.k"], kv[:, D:], sv["o2"], do, dqn, dkn, dkv[:, D:], B, self.H, M, N, self.scale,
                         sv["sm2"], sv["sl2"], zero_attn=getattr(xa, "allow_zero_attn", False), **xa_mask)
            self._qk_norm_bwd(xa, sv, "xqkn", dqn, dkn, sv["q"], kv[:, :D], dq, dkv[:, :D], Rq, Rc)
        else:
            ops.attn_bwd(sv["q"], kv[:, :D], kv[:, D:], sv["o2"], do, dq, dkv[:, :D], dkv[:, D:], B, self.H, M, N, self.scale,
                         sv["sm2"], sv["sl2"], zero_attn=getattr(xa, "allow_zero_attn", False), **xa_mask)
        self._dW(dq, sv["hq"], xa.q, Rq)
        dhq = ws.get("bwd.dh", (Rqp, D), bf)
        ops.gemm_nt(dq, self.wt(xa.q.weight), dhq, M=Rq, N=D, K=D)
        g_bf = ws.get("bwd.dec.gbf2", tuple(g_in.shape), g_in.dtype)
        self._ln_bwd(blk.query_norm, dhq, sv["y1"], sv, "nq", g, g_bf, Rq, dres=g)
        if dp:
            ops.scale_rows_bf16(g_bf, dp[0], M, Rq)
        if hoist is None:
            self._dW(dkv, sv["hc"], xa.kv, Rc)
            dhc = ws.get("bwd.dhc", (Rcp, D), bf)
            ops.gemm_nt(dkv, self.wt(xa.kv.weight), dhc, M=Rc, N=D, K=2 * D)
            self._ln_bwd(blk.context_norm, dhc, ctx, sv, "nc", dctx, dctx_bf, Rc, dres=dctx)     # accumulates over layers
        elif xa.kv.weight.requires_grad or blk.context_norm.weight.requires_grad:
            self._dw_jobs.append((dkv, hoist[1], hoist[2], 2 * D, D, Rc))                       # dL/d(W diag(gamma)) = dkv^T x_hat
        # self attention
        dh = self._self_attn_bwd(blk.self_attn, sv, g_bf, B, M, Rq, Rqp, sa_mask)
        self._flush_dW()
        self._ln_bwd(blk.norm1, dh, sv["y_in"], sv, "n1", g, g_in, Rq, dres=g)

    def _embed_bwd(self, sel, dx, dx_extra, is_dec):
        m = self.model
        embs = m.decoder_embeddings if is_dec else m.encoder_embeddings
        d = L.EmbedBwdDesc()
        for i, n in enumerate(sel["names"]):
            e, md = embs[n], d.mods[i]
            md.kind = e.kind
            if e.kind in (L.KIND_TOK, L.KIND_SEQ) and not (is_dec and e.kind == L.KIND_TOK):
                w = e.token_emb.weight
                md.d_table = self.grad_view(w).data_ptr() if w.requires_grad else None
                pad = e.token_emb.padding_idx
                md.has_padding_idx, md.padding_idx = (1, pad) if pad is not None else (0, 0)
            if isinstance(e.pos_emb, nn.Parameter) and e.pos_emb.requires_grad:
                md.d_pos = self.grad_view(e.pos_emb).data_ptr()
            if e.mod_emb.requires_grad:
                md.d_mod_emb = self.grad_view(e.mod_emb).data_ptr()
            if e.kind == L.KIND_SEQ_EMB and e.emb_proj.bias.requires_grad:
                md.d_proj_bias = self.grad_view(e.emb_proj.bias).data_ptr()
        d.dx, d.lddx = dx.data_ptr(), dx.stride(0)
        d.slot_mod, d.slot_src, d.slot_pos = sel["slot_mod"].data_ptr(), sel["slot_src"].data_ptr(), sel["slot_pos"].data_ptr()
        if is_dec and m.mask_token.requires_grad:
            d.d_mask_token = self.grad_view(m.mask_token).data_ptr()
        if not is_dec and m.num_register_tokens and m.register_tokens.requires_grad:
            d.d_reg_tokens = self.grad_view(m.register_tokens).data_ptr()
        d.n_mods, d.batch, d.dim, d.Nt, d.is_decoder = len(sel["names"]), sel["B"], self.D, sel["Nt"], 1 if is_dec else 0
        with ops._prof("embed_bwd"):
            L.check(L.embed_bwd(ops.C.byref(d), ops._stream()))
        if dx_extra is not None:
            # the context gradient reaches pos_emb / mod_emb only (context = proj(x) + encoder_emb, fm.py:679)
            d2 = L.EmbedBwdDesc()
            ops.C.memmove(ops.C.byref(d2), ops.C.byref(d), ops.C.sizeof(d))
            for i in range(d.n_mods):
                d2.mods[i].d_table = None
                d2.mods[i].d_proj_bias = None
            d2.d_reg_tokens = None
            d2.dx, d2.lddx = dx_extra.data_ptr(), dx_extra.stride(0)
            with ops._prof("embed_bwd"):
                L.check(L.embed_bwd(ops.C.byref(d2), ops._stream()))

    def train_backward(self, grad_scale: torch.Tensor):
        """Backward of the last ``train_forward``.  ``grad_scale``: 1-element fp32 device tensor holding
        d(objective)/d(loss) (loss scaling, 1/accum_iter, ...).  Accumulates into the flat gradient store."""
        c = self._ctx
        if c is None:
            raise RuntimeError("train_backward without a preceding training forward")
        self._ctx, self._dw_jobs = None, None
        if self.reducer is not None:
            self.reducer.begin()
            # CUs left free of the persistent GEMM grids for RCCL's kernels: only while an exchange can be in flight, i.e. from here to finish() -
            # the forward runs on every CU (with 16 of 256 reserved the dense GEMM family loses its whole-round tilings: +3.6 ms per 4M-B step
            # when the reservation covers the whole step, profiles/r06_reserved_cus.txt)
            if getattr(self.reducer, "reserved_cus", 0):
                L.lib.fm_set_reserved_cus(int(self.reducer.reserved_cus))
        m, ws = self.model, self.ws
        enc, dec, st, hs = c["enc"], c["dec"], c["st"], c["hs"]
        B, N, Mt, D = enc["B"], enc["Nt"], dec["Nt"], self.D
        bf, f32 = self.adt, torch.float32
        Rq, Rc = B * Mt, B * N
        Rqp, Rcp = st["y_final"].shape[0], st["x_final"].shape[0]
        # ---- heads: d(logits) in place, then dY (grouped NT) and dW_head (grouped TN) -----------------
        ops.cross_entropy(hs["logits"], hs["perm"], hs["tile_group"], dec["target_ids"].view(-1), hs["vocab_t"], hs["seg_start"],
                          hs["seg_count"], hs["n"], hs["maxV"], hs["row_loss"], hs["row_lse"], hs["head_loss"], hs["total"],
                          loss_type=hs["loss_type"], grad_scale=grad_scale, write_grad=True)
        dyp = ws.get("bwd.dyp", (hs["Rp"], D), bf)
        ops.gemm_nt_grouped(hs["logits"], hs["g_bwd"], hs["tile_group"], dyp, D, max_K=ru(hs["maxV"], 64))
        heads = [m.decoder_embeddings[h] for h in hs["heads"]]
        if any(h.to_logits.weight.requires_grad for h in heads):
            outs = [self.grad_view(h.to_logits.weight) if h.to_logits.weight.requires_grad else None for h in heads]
            key = tuple(o.data_ptr() if o is not None else 0 for o in outs)
            cache = getattr(self, "_head_tn_groups", None)
            if cache is None or cache[0] != key:       # no per-step host-to-device table upload (it would fence the stream)
                cache = self._head_tn_groups = (key, ops.make_groups(
                    [dict(out=o, N=(v if o is not None else 0)) for o, v in zip(outs, hs["vocabs"])], self.device))
            tn = cache[1]
            ops.gemm_tn_grouped(hs["logits"], hs["yp"], tn, hs["seg_start"], hs["seg_count"], hs["n"], hs["maxV"], hs["Rp"], D)
        g = ws.get("bwd.g_dec", (Rqp, D), f32)
        g_bf = ws.get("bwd.g_dec_bf", (Rqp, D), bf)
        self._ln_bwd(m.decoder_norm, dyp, st["y_final"], hs["sv"], "dn", g, g_bf, Rq, dres=None, dy_row_map=hs["r2p"])
        self._stage("heads")
        # ---- decoder ----------------------------------------------------------------------------------
        dctx = ws.get("bwd.dctx", (Rcp, D), f32)
        dctx_bf = ws.get("bwd.dctx_bf", (Rcp, D), bf)
        Ld = len(m.decoder)
        ctx_hat = st.get("ctx_hat")
        if ctx_hat is None:
            with ops._prof("fills"):
                dctx.zero_()
        else:       # context-norm hoist: the layers' d(kv) side by side, dL/d(W_l diag(gamma_l)) in a zeroed fp32 scratch
            dkv_all = ws.get("bwd.dkv_all", (Rcp, Ld * 2 * D), bf)
            dwp = ws.get("bwd.dwp", (Ld, 2 * D, D), f32)
            with ops._prof("fills"):
                dwp.zero_()
        for i in reversed(range(Ld)):
            sv = st["dec_layers"][i]
            if "ckpt_in" in sv:       # recompute this block's activations from its saved input (one shared set of buffers)
                y_in, dp, sv = sv["ckpt_in"], sv["dp"], {}
                self.decoder_block_fwd(m.decoder[i], y_in, st["ctx"], B, Mt, N, st["smask"], st["emask"], sv, "ckpt", out_name="ckpt.out", dp=dp,
                                       ctx_hat=ctx_hat)
            hoist = None if ctx_hat is None else (dkv_all[:, i * 2 * D:(i + 1) * 2 * D], ctx_hat, dwp[i])
            self.decoder_block_bwd(m.decoder[i], sv, g, g_bf, dctx, dctx_bf, st["ctx"], B, Mt, N, st["smask"], st["emask"], hoist=hoist)
            self._stage(f"dec{i}")
        if ctx_hat is not None and Ld:
            # d(x_hat) = sum_l dkv_l (W_l diag(gamma_l)): one GEMM with a reduction over all layers, then ONE LayerNorm backward (no affine part)
            dxh = ws.get("bwd.dhc", (Rcp, D), bf)
            ops.gemm_nt(dkv_all, self.wt_fold_stack(), dxh, M=Rc, N=D, K=Ld * 2 * D)
            self._ln_bwd(self._unit_norm(), dxh, st["ctx"], st["sv_hat"], "ch", dctx, dctx_bf, Rc, dres=None)
            # dL/dW_l = dL/dW'_l diag(gamma_l);  dL/dgamma_l = column sums of dL/dW'_l * W_l
            jobs = [(dwp[i], b.cross_attn.kv.weight.detach(), b.context_norm.weight.detach(), self._g(b.cross_attn.kv.weight), self._g(b.context_norm.weight))
                    for i, b in enumerate(m.decoder) if b.cross_attn.kv.weight.requires_grad or b.context_norm.weight.requires_grad]
            if jobs:
                ops.fold_colscale_grad(jobs)
        self._embed_bwd(dec, g, None, True)
        self._stage("dec_emb")
        # ---- context projection + encoder -------------------------------------------------------------
        top = st["top"]
        if len(m.decoder) == 0:
            ops.f32_to_bf16(dctx, dctx_bf)
        self._dW(dctx_bf, top["xn"], m.decoder_proj_context, Rc)
        dxn = ws.get("bwd.dh", (Rcp, D), bf)
        ops.gemm_nt(dctx_bf, self.wt(m.decoder_proj_context.weight), dxn, M=Rc, N=D, K=D)
        ge = ws.get("bwd.g_enc", (Rcp, D), f32)
        ge_bf = ws.get("bwd.g_enc_bf", (Rcp, D), bf)
        self._ln_bwd(m.encoder_norm, dxn, st["x_final"], top, "en", ge, ge_bf, Rc, dres=None)
        self._stage("ctx")
        for i in reversed(range(len(m.encoder))):
            sv = st["enc_layers"][i]
            if "ckpt_in" in sv:
                x_in, dp, sv = sv["ckpt_in"], sv["dp"], {}
                self.encoder_block_fwd(m.encoder[i], x_in, B, N, st["emask"], sv, "ckpt", out_name="ckpt.out", dp=dp)
            self.encoder_block_bwd(m.encoder[i], sv, ge, ge_bf, B, N, st["emask"])
            self._stage(f"enc{i}")
        # d(x0) -> token tables / projections / embeddings; d(ctx) also reaches the encoder embeddings
        for n in enc["names"]:
            e = m.encoder_embeddings[n]
            if e.kind == L.KIND_PATCH and e.proj.weight.requires_grad:
                ops.gemm_tn(ge_bf, enc["patch_rows"], self.grad_view(e.proj.weight), N=D, K=e.proj.weight.shape[1], R=Rc)
            elif e.kind == L.KIND_SEQ_EMB and e.emb_proj.weight.requires_grad:
                ops.gemm_tn(ge_bf, enc["seqemb_rows"], self.grad_view(e.emb_proj.weight), N=D, K=e.emb_proj.weight.shape[1], R=Rc)
        self._embed_bwd(enc, ge, dctx, False)
        if self.reducer is not None:
            self.reducer.finish()        # remaining slices + wait: gradients are averaged when backward returns
            if getattr(self.reducer, "reserved_cus", 0) and not DP_RESERVE_ALWAYS:
                L.lib.fm_set_reserved_cus(0)
