"""Optimizer construction with upstream's parameter grouping (``fourm/utils/optim_factory.py:111-244``)
and a fused AdamW whose state is indistinguishable from ``torch.optim.AdamW``'s.

FusedAdamW runs one HIP kernel per *contiguous run* of parameters that share a group: the engine lays
parameters and gradients out in flat fp32 buffers, so a 360-tensor model updates in a handful of
launches, each a single streaming pass over (param, grad, exp_avg, exp_avg_sq)."""
import json

import torch
from torch import optim


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None, decoder_decay=None,
                         decoder_list=(), no_lr_scale_list=()):
    """norm / bias / skip-listed tensors get no decay; optional per-layer lr scale.  Group dicts carry
    'lr_scale' (consumed by the trainer's schedule, run_training_4m.py:707-711)."""
    names, groups = {}, {}
    for name, p in model.named_parameters():
        name = name.replace("_fsdp_wrapped_module.", "")
        if not p.requires_grad:
            continue
        if ("norm." in name or ".norm" in name or name.endswith(".bias") or name.endswith(".lookup_table_weight")
                or name.endswith(".gamma") or name in skip_list):
            gname, wd = "no_decay", 0.
        elif decoder_decay is not None and (name.startswith("decoder.") or name in decoder_list):
            gname, wd = "decoder_decay", decoder_decay
        else:
            gname, wd = "decay", weight_decay
        skip_scale = False
        layer_id = None
        if get_num_layer is not None:
            layer_id = get_num_layer(name)
            gname = "layer_%d_%s" % (layer_id, gname)
            if name in no_lr_scale_list:
                skip_scale = True
                gname = f"{gname}_no_lr_scale"
        if gname not in groups:
            scale = get_layer_scale(layer_id) if (get_layer_scale is not None and not skip_scale) else 1.
            names[gname] = {"weight_decay": wd, "params": [], "lr_scale": scale}
            groups[gname] = {"weight_decay": wd, "params": [], "lr_scale": scale}
        groups[gname]["params"].append(p)
        names[gname]["params"].append(name)
    print("Param groups = %s" % json.dumps(names, indent=2))
    return list(groups.values())


class FusedAdamW(optim.AdamW):
    """torch.optim.AdamW with the update executed by ``fm_adamw`` (csrc/elementwise.hip).

    ``state_dict()`` / ``load_state_dict()`` are the parent's: per-parameter 'step' (fp32 scalar tensor),
    'exp_avg', 'exp_avg_sq'; param_groups keep their extra keys ('lr_scale').  amsgrad / maximize are
    not supported."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, **kw):
        if kw.get("amsgrad") or kw.get("maximize"):
            raise NotImplementedError("FusedAdamW: amsgrad / maximize are not implemented")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        for g in self.param_groups:
            for p in g["params"]:
                # fm_adamw reinterprets the storage as float*: anything else would be silently corrupted
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                    raise TypeError(f"FusedAdamW handles contiguous fp32 device parameters only (got {p.dtype}, "
                                    f"{'contiguous' if p.is_contiguous() else 'strided'}, {p.device}); use torch.optim.AdamW")
        self._clip_coef = None        # device scalar set by fused_grad_norm(clip=...)
        self._runs = None
        self._shadow_tables = {}      # (group index, step) signature -> cached device job table of fm_adamw_shadow
        # hipGraph support (fourm.hip.graph.GraphedTrainStep): per-group device block {lr, wd, 1 - b1^t, sqrt(1 - b2^t)} that the
        # kernels read instead of their scalar arguments; in captured mode step() launches only, the host bookkeeping (step counts,
        # hyper upload) is done by advance_host_state() before each replay
        self._hyper_dev = self._hyper_host = None
        self._captured = False

    @staticmethod
    def _check_grad(p):
        g = p.grad
        if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
            raise TypeError(f"FusedAdamW: gradient of a {tuple(p.shape)} parameter is {g.dtype} / "
                            f"{'contiguous' if g.is_contiguous() else 'strided'} on {g.device}; fp32 contiguous expected")

    # -- hipGraph support -----------------------------------------------------------------------
    def enable_device_hyper(self, device, slots: int = 8):
        """Device block + a RING of pinned staging buffers: ``advance_host_state`` only enqueues an asynchronous copy, and the host
        may run several replays ahead of the GPU - a single staging buffer would be overwritten with step k+1's values before step
        k's copy has executed.  A slot is reused only after the copy that last read it has run (event), i.e. the host blocks only
        when it is more than ``slots`` steps ahead."""
        n = len(self.param_groups)
        self._hyper_dev = torch.zeros(n, 4, dtype=torch.float32, device=device)
        self._hyper_ring = [torch.zeros(n, 4, dtype=torch.float32).pin_memory() for _ in range(slots)]
        self._hyper_events = [None] * slots
        self._hyper_slot = 0
        self._hyper_host = self._hyper_ring[0]          # (the values of the most recent step; kept for introspection)

    def advance_host_state(self):
        """What step() does on the host, for a step whose launches are replayed from a graph: step counters + the hyper block
        (one 16-byte-per-group asynchronous copy on the current stream, from the next slot of the staging ring)."""
        if self._hyper_dev is None:
            raise RuntimeError("enable_device_hyper() first")
        i = self._hyper_slot
        if self._hyper_events[i] is not None:
            self._hyper_events[i].synchronize()         # the copy that last used this staging buffer has executed
        host = self._hyper_ring[i]
        host.copy_(self._hyper_host)                     # groups without a step this time keep their previous values
        for gi, g in enumerate(self.param_groups):
            step = None
            for p in g["params"]:
                st = self.state.get(p)
                if p.grad is not None and st:
                    st["step"] += 1
                    step = int(st["step"])
            if step is None:
                continue
            b1, b2 = g["betas"]
            host[gi, 0], host[gi, 1] = g["lr"], g["weight_decay"]
            host[gi, 2], host[gi, 3] = 1.0 - b1 ** step, (1.0 - b2 ** step) ** 0.5
        self._hyper_dev.copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._hyper_events[i] = ev
        self._hyper_host = host
        self._hyper_slot = (i + 1) % len(self._hyper_ring)

    def _hyper(self, gi):
        """Device hyper block of group ``gi`` while a graph is being captured (eager steps pass the scalars themselves)."""
        return self._hyper_dev[gi] if (self._captured and self._hyper_dev is not None) else None

    # -- contiguous runs ------------------------------------------------------------------------
    def _shadowed(self):
        """Weight matrices whose bf16 GEMM-operand copies (engine shadows) can be rewritten by the update itself:
        {group index: [(param, plain dst | None, transposed dst | None, [(engine, shadow key)])]}."""
        out = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                regs = getattr(p, "_fourm_shadows", None)
                if p.grad is None or not regs or p.dim() < 2:
                    continue
                plain = tr = None
                keys = []
                for ref, key in regs:
                    eng = ref()
                    sh = eng.shadows.get(key) if eng is not None else None
                    if sh is None:
                        continue
                    for job in sh.jobs:
                        pp, dst, transposed = job[:3]
                        # (transposed copies keep going through the engine's lazy fm_shadow_refresh: a transposing walk cannot
                        # stream, and this kernel must; so do images with a folded LayerNorm weight: they depend on a second parameter)
                        if len(job) > 3 and job[3] is not None:
                            continue
                        if pp is p and not transposed and plain is None:
                            plain = dst; keys.append((eng, key))
                if plain is not None or tr is not None:
                    out.setdefault(gi, []).append((p, plain, tr, keys))
        return out

    def _step_shadowed(self, shadowed):
        """One fm_adamw_shadow launch per (group, step count): master update + bf16 copies in one pass."""
        from fourm.hip import ops
        done, touched, written = set(), [], set()
        for gi, items in shadowed.items():
            g = self.param_groups[gi]
            by_step = {}
            for it in items:
                p = it[0]
                self._check_grad(p)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                if not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous() and st["exp_avg"].dtype == torch.float32):
                    continue                      # odd state (foreign checkpoint): the plain path handles it
                if not self._captured:
                    st["step"] += 1
                by_step.setdefault(int(st["step"]), []).append(it)
                done.add(id(p))
            for step, its in by_step.items():
                sig = (gi,) + tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr(),
                                     pl.data_ptr() if pl is not None else 0, tr.data_ptr() if tr is not None else 0) for p, pl, tr, _ in its)
                cached = self._shadow_tables.get(gi)
                if cached is None or cached[0] != sig:       # rebuilt only when a buffer moved: no per-step host-to-device upload
                    jobs = [dict(p=p, g=p.grad, m=self.state[p]["exp_avg"], v=self.state[p]["exp_avg_sq"], plain=pl, t=tr) for p, pl, tr, _ in its]
                    table, tiles = ops.adamw_jobs_table(jobs, its[0][0].device)
                    cached = self._shadow_tables[gi] = (sig, table, len(jobs), tiles)
                _, table, n, tiles = cached
                ops.adamw_shadow(table, n, tiles, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], max(step, 1), self._clip_coef,
                                 hyper=self._hyper(gi), sumsq=getattr(self, "_ss", None))
                for p, pl, tr, keys in its:
                    touched += keys
                    for dst in (pl, tr):
                        if dst is not None:
                            written.add((id(p), dst.data_ptr()))
        return done, touched, written

    def _build_runs(self, skip=()):
        """Per group: maximal runs of parameters adjacent in memory (flat parameter store) whose gradients
        are adjacent with the same spacing.  State tensors of a run are carved from one flat buffer."""
        runs = []
        for gi, g in enumerate(self.param_groups):
            ps = [p for p in g["params"] if p.grad is not None and id(p) not in skip]
            for p in ps:
                self._check_grad(p)
            ps.sort(key=lambda p: p.data_ptr())
            cur = []
            for p in ps:
                if cur:
                    q = cur[-1]
                    # adjacent AND carved from the same allocation (two autograd-allocated tensors that merely sit
                    # back to back in the caching allocator are separate storages: a merged view would be out of bounds)
                    adj = (p.data_ptr() == q.data_ptr() + q.numel() * 4 and p.grad.data_ptr() == q.grad.data_ptr() + q.numel() * 4
                           and _same_storage(p, q) and _same_storage(p.grad, q.grad) and self._state_adjacent(q, p))
                    if not adj:
                        runs.append((gi, cur))
                        cur = []
                cur.append(p)
            if cur:
                runs.append((gi, cur))
        return runs

    def _state_adjacent(self, q, p):
        sq, sp = self.state.get(q), self.state.get(p)
        if not sq and not sp:
            return True          # both uninitialised: will be carved adjacently
        if not sq or not sp:
            return False
        return (sp["exp_avg"].data_ptr() == sq["exp_avg"].data_ptr() + q.numel() * 4
                and sp["exp_avg_sq"].data_ptr() == sq["exp_avg_sq"].data_ptr() + q.numel() * 4
                and _same_storage(sp["exp_avg"], sq["exp_avg"]) and _same_storage(sp["exp_avg_sq"], sq["exp_avg_sq"])
                and float(sp["step"]) == float(sq["step"]))

    def _init_state(self, run):
        need = [p for p in run if len(self.state[p]) == 0]
        if not need:
            return
        n = sum(p.numel() for p in run)
        dev = run[0].device
        m, v = torch.zeros(n, device=dev, dtype=torch.float32), torch.zeros(n, device=dev, dtype=torch.float32)
        o = 0
        for p in run:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = m[o:o + p.numel()].view_as(p)
                st["exp_avg_sq"] = v[o:o + p.numel()].view_as(p)
            o += p.numel()

    @torch.no_grad()
    def fused_grad_norm(self, clip=None, lazy=False):
        """L2 norm of every gradient, computed on the device; with ``clip`` the coefficient
        min(1, clip / (norm + 1e-6)) is folded into the next step().  Returns a device scalar.
        ``lazy`` (no clipping only): the norm is not needed to take the step, so the sum of squares rides on the AdamW kernels' own pass over the
        gradients (fm_adamw*'s ``sumsq``) instead of a separate 4 B/param read: the returned scalar is FILLED BY THE NEXT step() - read it
        after that call (NativeScaler returns it to the trainer after optimizer.step(), like upstream's get_grad_norm_ value)."""
        from fourm.hip import ops
        self._lazy_norm = None
        if lazy and clip is None:
            dev = next((p.grad.device for g in self.param_groups for p in g["params"] if p.grad is not None), None)
            if dev is None:
                return torch.tensor(0.)
            ss, norm = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
            self._lazy_norm, self._clip_coef = (ss, norm), None
            return norm[0]
        grads = []
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is not None:
                    self._check_grad(p)
                    grads.append(p.grad)
        if not grads:
            return torch.tensor(0.)
        dev = grads[0].device
        ss, norm = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        coef = torch.zeros(1, device=dev) if clip is not None else None
        seen = set()
        # merge adjacent gradient tensors so that a flat gradient buffer is one launch
        grads.sort(key=lambda t: t.data_ptr())
        start, n = None, 0
        for t in grads:
            if t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            if start is not None and t.data_ptr() == start.data_ptr() + n * 4 and _same_storage(t, start):
                n += t.numel()
                continue
            if start is not None:
                ops.sumsq(_flat_view(start, n), ss)
            start, n = t, t.numel()
        ops.sumsq(_flat_view(start, n), ss)
        ops.clip_coef(ss, clip if clip is not None else 0.0, norm, coef)
        self._clip_coef = coef
        return norm[0]

    @torch.no_grad()
    def step(self, closure=None):
        from fourm.hip import engine, ops
        loss = closure() if closure is not None else None
        lazy = getattr(self, "_lazy_norm", None)
        self._ss = lazy[0] if lazy is not None else None          # device scalar the AdamW launches of this step add sum g^2 to
        done, touched, written = self._step_shadowed(self._shadowed())
        for gi, run in self._build_runs(skip=done):
            g = self.param_groups[gi]
            self._init_state(run)
            st0 = self.state[run[0]]
            if not self._captured:
                for p in run:
                    self.state[p]["step"] += 1
            step = max(int(st0["step"]), 1)
            n = sum(p.numel() for p in run)
            if not self._state_adjacent_all(run):
                for p in run:       # state loaded from a checkpoint (separate tensors): per-tensor launches
                    st = self.state[p]
                    ops.adamw(p, p.grad, st["exp_avg"], st["exp_avg_sq"], p.numel(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                              g["weight_decay"], max(int(st["step"]), 1), self._clip_coef, hyper=self._hyper(gi), sumsq=self._ss)
                continue
            ops.adamw(run[0], run[0].grad, st0["exp_avg"], st0["exp_avg_sq"], n, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                      g["weight_decay"], step, self._clip_coef, hyper=self._hyper(gi), sumsq=self._ss)
        if lazy is not None:
            ops.clip_coef(lazy[0], 0.0, lazy[1], None)              # norm = sqrt(sum g^2): the scalar fused_grad_norm(lazy=True) returned
            self._lazy_norm = self._ss = None
        self._clip_coef = None
        engine.bump_weight_epoch()
        for eng in {id(e): e for e, _ in touched}.values():          # the copies written above are current again
            eng.mark_shadows_fresh([k for e, k in touched if e is eng], written)
        return loss

    def _state_adjacent_all(self, run):
        return all(self._state_adjacent(a, b) for a, b in zip(run[:-1], run[1:]))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():      # keep the layout torch uses for non-capturable AdamW
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().float().cpu()


def _same_storage(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()


def _flat_view(t: torch.Tensor, n: int) -> torch.Tensor:
    """n elements starting at t's first one; the callers only merge tensors of ONE storage, so the view is in bounds."""
    return torch.as_strided(t, (n,), (1,))


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    opt_lower = args.opt.lower()
    weight_decay = args.weight_decay
    decoder_decay = getattr(args, "decoder_decay", None)
    no_lr_scale_list = args.no_lr_scale_list.split("-") if getattr(args, "no_lr_scale_list", None) else []
    if weight_decay and filter_bias_and_bn:
        skip = skip_list if skip_list is not None else (model.no_weight_decay() if hasattr(model, "no_weight_decay") else {})
        decoder = model.decoder_weight_decay() if hasattr(model, "decoder_weight_decay") else {}
        parameters = get_parameter_groups(model, weight_decay, skip, get_num_layer, get_layer_scale, decoder_decay, decoder, no_lr_scale_list)
        weight_decay = 0.
    else:
        parameters = model.parameters()
    opt_args = dict(lr=args.lr, weight_decay=weight_decay)
    if getattr(args, "opt_eps", None) is not None:
        opt_args["eps"] = args.opt_eps
    if getattr(args, "opt_betas", None) is not None:
        opt_args["betas"] = tuple(args.opt_betas)
    print("optimizer settings:", opt_args)
    kind = opt_lower.split("_")[-1]
    if kind in ("sgd", "nesterov"):
        opt_args.pop("eps", None)
        return optim.SGD(parameters, momentum=args.momentum, nesterov=True, **opt_args)
    if kind == "momentum":
        opt_args.pop("eps", None)
        return optim.SGD(parameters, momentum=args.momentum, nesterov=False, **opt_args)
    if kind == "adam":
        return optim.Adam(parameters, **opt_args)
    if kind == "adamw":
        # the fused kernel is for engine-backed models (flat fp32 parameter / gradient stores); anything else gets torch's AdamW
        inner = getattr(model, "module", model)
        ps = list(model.parameters())
        fused_ok = hasattr(type(inner), "engine") and ps and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in ps)
        return FusedAdamW(parameters, **opt_args) if fused_ok else optim.AdamW(parameters, **opt_args)
    raise ValueError(f"Invalid optimizer {args.opt}")


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
