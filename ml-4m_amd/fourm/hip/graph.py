"""The whole train step as ONE hipGraph: select + embed -> encoder -> decoder -> heads / cross-entropy -> hand-written backward ->
gradient norm (+ clipping coefficient) -> fused AdamW, ~1500 kernel launches captured once and replayed with one host call.

Why it is legal here: every shape of a step is static, ragged per-modality counts live in device-side tables, nothing in the step
synchronises with the host, the workspace is allocated once, and the optimizer's per-step scalars (learning rate, weight decay,
bias corrections) are read from a 16-byte device block per parameter group that the host refreshes before each replay.

What is frozen at capture time (documented deviations from the eager path):
  * the batch tensors are STATIC buffers: ``step(mod_dict)`` copies the new batch into them (same shapes / dtypes required);
  * the decoder modality order (upstream re-draws it with ``random.sample`` at every forward, fm.py:306) is the one drawn at capture;
    with ``decoder_sep_mask`` the loss does not depend on it;
  * the parameter set with gradients, loss type, token budgets, clipping mode.
A graphed step refuses to run (loudly) if the engine's workspace or flat stores were re-allocated after capture.

Data parallel: with a gradient reducer that goes through torch.distributed the step stays eager (torch's NCCL watchdog polls the
captured collectives' events: hipErrorCapturedEvent, round 4).  With the DIRECT exchange (``DataParallel(comm="direct")``: ncclAllReduce
through ctypes on a side stream, event fences, fourm/parallel/rccl.py) the collectives are nodes of the graph: pass the wrapper as
``data_parallel``.  Exercised at world size 1 with forced collectives (tests/test_parallel_gpu.py); no N > 1 run exists.
"""
import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, mod_dict, num_encoder_tokens: int, num_decoder_tokens: int, loss_type: str = "mod",
                 clip_grad=None, warmup: int = 2, order_seed=None, data_parallel=None):
        """``order_seed``: seed ``random`` right before the captured forward, i.e. choose the (frozen) decoder modality order."""
        from fourm.utils.optim_factory import FusedAdamW
        if not isinstance(optimizer, FusedAdamW):
            raise TypeError("GraphedTrainStep needs FusedAdamW (device-side hyper-parameters)")
        eng = model.engine
        self.dp = data_parallel
        if data_parallel is not None:
            if getattr(data_parallel, "_comm", "torch") != "direct":
                raise RuntimeError("GraphedTrainStep(data_parallel=...) needs DataParallel(comm='direct'): collectives issued through torch.distributed "
                                   "cannot be captured (torch's NCCL watchdog polls their events)")
            data_parallel.time_exchange = False          # (timing events are not recorded inside a capture)
        elif eng.reducer is not None:
            raise RuntimeError("a gradient reducer is attached (data parallel): pass the DataParallel wrapper as data_parallel (comm='direct'), or stay eager")
        self.model, self.opt, self.n_enc, self.n_dec, self.loss_type, self.clip = model, optimizer, num_encoder_tokens, num_decoder_tokens, loss_type, clip_grad
        dev = model.mask_token.device
        self.static = {m: {k: v.detach().clone().to(dev) for k, v in d.items() if torch.is_tensor(v)} for m, d in mod_dict.items()}
        if optimizer._hyper_dev is None:
            optimizer.enable_device_hyper(dev)
        # eager warm-up: allocates the workspace, weight shadows, optimizer state and every cached device table (two steps at least:
        # the table of "every shadow stale after an optimizer step" is first built by the SECOND forward)
        for _ in range(max(2, warmup)):
            self._eager_step()
        if data_parallel is not None and getattr(getattr(eng, "reducer", None), "_direct", None) is None and getattr(eng.reducer, "world", 1) > 1:
            # DataParallel falls back to torch.distributed collectives when the direct binding cannot be honoured (store not on a GPU, backend
            # not nccl): capturing those is the hipErrorCapturedEvent crash this path exists to rule out
            raise RuntimeError("GraphedTrainStep: the gradient reducer did not get its direct RCCL communicator (comm='direct' was downgraded); stay eager")
        self._sig = self._signature()
        torch.cuda.synchronize()
        optimizer.zero_grad(set_to_none=True)           # the captured backward starts a fresh accumulation window (memset captured)
        optimizer._captured = True
        self.graph = torch.cuda.CUDAGraph()
        if order_seed is not None:
            import random
            random.seed(order_seed)
        try:
            with torch.cuda.graph(self.graph):
                self.loss, self.mod_loss, self.norm = self._launches()
        finally:
            optimizer._captured = False
        # the capture itself executed nothing; host-side state that step() would have advanced is advanced per replay
        self.steps = 0

    def _signature(self):
        eng = self.model.engine
        return (eng.flat_params.data_ptr(), eng.flat_grads.data_ptr(), len(eng.ws.bufs), tuple(sorted((k, v.data_ptr()) for k, v in eng.ws.bufs.items()))[:8])

    def _launches(self):
        loss, mod_loss = (self.dp or self.model)(self.static, self.n_enc, self.n_dec, loss_type=self.loss_type)
        loss.backward()
        norm = self.opt.fused_grad_norm(clip=self.clip, lazy=True)
        self.opt.step()
        return loss.detach(), {k: v.detach() for k, v in mod_loss.items()}, norm

    def _eager_step(self):
        out = self._launches()
        self.opt.zero_grad(set_to_none=True)
        return out

    def resync(self):
        """After the weights were changed OUTSIDE the graph (checkpoint load, manual edits): rebuild every bf16 weight shadow eagerly.
        (The captured forward only refreshes the copies the captured optimizer step leaves stale: the transposed ones.)"""
        from fourm.hip import engine as E
        E.bump_weight_epoch()
        self.model.engine._refresh_shadows()

    def step(self, mod_dict=None):
        """Copy the batch into the static buffers (when given), refresh the optimizer's device scalars, replay.
        Returns (loss, {mod: loss}, grad_norm) as device tensors that are overwritten by the next replay."""
        if self._signature() != self._sig:
            raise RuntimeError("the engine's workspace or flat parameter / gradient stores moved after capture: build a new GraphedTrainStep")
        if mod_dict is not None:
            for m, d in self.static.items():
                for k, v in d.items():
                    src = mod_dict[m][k]
                    if src.shape != v.shape or src.dtype != v.dtype:
                        raise ValueError(f"{m}.{k}: {tuple(src.shape)} {src.dtype} does not match the captured {tuple(v.shape)} {v.dtype}")
                    v.copy_(src, non_blocking=True)
        self.opt.advance_host_state()
        self.graph.replay()
        self.steps += 1
        return self.loss, self.mod_loss, self.norm
