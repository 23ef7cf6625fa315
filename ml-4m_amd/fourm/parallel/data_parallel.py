"""Pure data parallelism over the GPUs of one node: one process per GPU, weights replicated, one
exchange per optimizer step — the mean of the fp32 gradients — replacing upstream's
``DistributedDataParallel`` (run_training_4m.py:512).

MI355X specifics that shape the implementation
  * gradients live in ONE flat fp32 buffer (fourm.hip.engine), so a bucket is a contiguous slice: no
    flatten / unflatten copies, and few, large RCCL calls (xGMI is point-to-point, 7 links x ~153 GB/s
    per GPU: large messages are what reaches link bandwidth);
  * the hand-written backward reports *stages* (heads, each decoder block, decoder embeddings, context
    projection, each encoder block, encoder embeddings).  When a stage completes, the slices whose
    gradients became final are all-reduced on RCCL's own stream while the next stage computes;
  * ``ReduceOp.AVG`` averages inside the collective (no separate scaling pass).

``GradReducer`` holds the bucketing / overlap logic and is backend-agnostic (tested with gloo on CPU).
"""
import contextlib
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn


class GradReducer:
    """Mean of slices of a flat gradient buffer across the ranks, launched stage by stage.

    ranges_by_stage: {stage: [(offset, length), ...]} in elements of ``flat``.  Adjacent ranges are
    merged; ranges larger than ``bucket_elems`` are split so that several collectives are in flight.

    algorithm  "all_reduce" (default) | "reduce_scatter": reduce-scatter + all-gather of each bucket (the bucket's head of
               length numel // world * world; the tail goes through all_reduce) - the same bytes as a ring all-reduce, issued as
               the two collectives a sharded optimizer would split apart.
    wire_dtype None (fp32 on the wire) | torch.bfloat16: the bucket is packed to bf16 on the compute stream, exchanged, and unpacked
               (x 1/world) into the fp32 store; accumulation ACROSS ranks is then in bf16 (upstream's FSDP script reduces in bf16
               too, run_training_4m_fsdp.py:528), half the xGMI bytes.
    force      run the collectives even at world size 1 (tests of the RCCL code path on one GPU)."""

    def __init__(self, flat: torch.Tensor, ranges_by_stage: Dict[str, Sequence[Tuple[int, int]]], group=None,
                 bucket_elems: int = 64 * 1024 * 1024, algorithm: str = "all_reduce", wire_dtype=None, force: bool = False,
                 min_launch_mb: float = 0.0, pack=None, unpack=None, comm: str = "torch"):
        """``pack(src_f32, dst_wire)`` / ``unpack(src_wire, dst_f32, scale)``: the wire-format conversions; default = the HIP kernels
        fm_f32_to_bf16 / fm_bf16_to_f32_scaled on the compute stream (the CPU tests of the bucketing logic inject torch copies: there is
        no other implementation behind the default)."""
        if algorithm not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"algorithm {algorithm!r}")
        if wire_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("wire_dtype: None / torch.float32 / torch.bfloat16")
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = bucket_elems
        self.algorithm, self.force = algorithm, force
        self.wire = torch.bfloat16 if wire_dtype == torch.bfloat16 else None
        self.stages = {s: self._merge(r) for s, r in ranges_by_stage.items()}
        covered = sorted((o, n) for r in self.stages.values() for o, n in r)
        for (o1, n1), (o2, _) in zip(covered, covered[1:]):
            if o1 + n1 > o2:
                raise ValueError("gradient ranges of different stages overlap")
        self._pending = []
        self._done = set()
        self._waiting = []
        self.min_launch_elems = (1 << 62) if min_launch_mb == float("inf") else int(min_launch_mb * 1024 * 1024 // 4)
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"          # RCCL averages in the collective; gloo has no AVG
        self._pack, self._unpack = pack, unpack
        # comm = "direct": RCCL called on a side stream with event fences (fourm.parallel.rccl), no torch Work objects - the form a captured
        # train step needs (torch's NCCL watchdog polls captured events: DESIGN §6).  GPU stores only; "torch" is the default.
        if comm not in ("torch", "direct"):
            raise ValueError(f"comm {comm!r}: 'torch' or 'direct'")
        self._direct = None
        if comm == "direct" and (self.world > 1 or force):
            if not flat.is_cuda:
                raise ValueError("comm='direct' needs the gradient store on a GPU")
            from .rccl import shared_comm
            self._direct = shared_comm(group, flat.device)          # (one communicator per group and device, whatever the number of reducer rebuilds)
            self._avg = True
        self._wire_bufs = {}                   # (offset, length) -> bf16 staging buffer, allocated once
        self._shards = {}
        # measurement hooks (bench.py): events around finish() on the compute stream = the part of the exchange the backward did not hide
        self.time_exchange = False
        self._exch_events = []
        self.n_collectives = 0                 # collectives launched since begin()
        self.bytes_on_wire = 0                 # payload handed to them since begin()

    def _merge(self, ranges):
        out = []
        for o, n in sorted(ranges):
            if n <= 0:
                continue
            if out and out[-1][0] + out[-1][1] == o:
                out[-1] = (out[-1][0], out[-1][1] + n)
            else:
                out.append((o, n))
        split = []
        for o, n in out:
            while n > self.bucket_elems:
                split.append((o, self.bucket_elems))
                o, n = o + self.bucket_elems, n - self.bucket_elems
            split.append((o, n))
        return split

    def begin(self):
        self._pending, self._done, self._waiting = [], set(), []
        self.n_collectives, self.bytes_on_wire = 0, 0

    def exposed_ms(self, clear: bool = True):
        """Per finish() call: milliseconds the compute stream spent between entering finish() and having every reduced slice back
        (waiting for collectives that were still running + launching / running the ones that had not started + the unpack kernels).
        Synchronises.  Only recorded while ``time_exchange`` is set and the store lives on a GPU."""
        out = []
        for e0, e1 in self._exch_events:
            e1.synchronize()
            out.append(e0.elapsed_time(e1))
        if clear:
            self._exch_events = []
        return out

    def _exchange_direct(self, t: torch.Tensor, op):
        from .rccl import _DirectWork
        d = self._direct
        d.stream.wait_stream(torch.cuda.current_stream())               # the slice is final on the compute stream
        avg = op == dist.ReduceOp.AVG
        if self.algorithm == "reduce_scatter" and t.numel() >= self.world:
            head = t.numel() // self.world * self.world
            key = (t.data_ptr(), head)
            shard = self._shards.get(key)
            if shard is None:
                shard = self._shards[key] = torch.empty(head // self.world, dtype=t.dtype, device=t.device)
            d.reduce_scatter(shard, t[:head], avg)
            d.all_gather(t[:head], shard)
            self.n_collectives += 2
            if head < t.numel():
                d.all_reduce_(t[head:], avg)
                self.n_collectives += 1
        else:
            d.all_reduce_(t, avg)
            self.n_collectives += 1
        self.bytes_on_wire += t.numel() * t.element_size()
        return [_DirectWork(d)]

    def _exchange(self, t: torch.Tensor, op):
        """Asynchronous collectives that leave the cross-rank reduction of ``t`` in ``t``; returns the work handles."""
        if self._direct is not None:
            return self._exchange_direct(t, op)
        if self.algorithm == "reduce_scatter" and t.numel() >= self.world:
            head = t.numel() // self.world * self.world
            key = (t.data_ptr(), head)
            shard = self._shards.get(key)
            if shard is None:
                shard = self._shards[key] = torch.empty(head // self.world, dtype=t.dtype, device=t.device)
            self.n_collectives += 2 + (head < t.numel())
            self.bytes_on_wire += t.numel() * t.element_size()
            works = [dist.reduce_scatter_tensor(shard, t[:head], op=op, group=self.group, async_op=True)]
            if not self._avg:
                works[0].wait()       # gloo runs asynchronous collectives on worker threads without ordering them (the world-size-4 test read an
                                      # unwritten shard); RCCL enqueues both on one stream, in order
            works.append(dist.all_gather_into_tensor(t[:head], shard, group=self.group, async_op=True))
            if head < t.numel():
                works.append(dist.all_reduce(t[head:], op=op, group=self.group, async_op=True))
            return works
        self.n_collectives += 1
        self.bytes_on_wire += t.numel() * t.element_size()
        return [dist.all_reduce(t, op=op, group=self.group, async_op=True)]

    def stage_done(self, stage: str, flush: bool = False):
        """Gradients of ``stage`` are final on the compute stream: start their exchange - once at least ``min_launch_elems`` elements
        are waiting (a collective costs the host ~50-100 us whatever its size: ~28 stages x 1-3 ranges per 4M-B step would be +3 ms of
        launch-side time, tools/overlap_dp.py; a few large messages are also what point-to-point xGMI links want)."""
        if (self.world == 1 and not self.force) or stage in self._done or stage not in self.stages:
            return
        self._done.add(stage)
        self._waiting = getattr(self, "_waiting", []) + list(self.stages[stage])
        if not flush and sum(n for _, n in self._waiting) < self.min_launch_elems:
            return
        ranges, self._waiting = self._merge(self._waiting), []
        for o, n in ranges:
            t = self.flat[o:o + n]
            if self.wire is not None:
                from fourm.hip import ops
                w = self._wire_bufs.get((o, n))
                if w is None:
                    w = self._wire_bufs[(o, n)] = torch.empty(n, dtype=self.wire, device=t.device)
                (self._pack or ops.f32_to_bf16)(t, w)                   # pack on the compute stream, behind the gradient kernels
                self._pending.append((self._exchange(w, dist.ReduceOp.SUM), t, w))
            else:
                op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
                self._pending.append((self._exchange(t, op), t, None))

    def finish(self):
        """Launch whatever stage has not been reported, then wait for every collective."""
        timed = self.time_exchange and self.flat.is_cuda and (self.world > 1 or self.force)
        if timed:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for s in self.stages:
            self.stage_done(s)
        if getattr(self, "_waiting", None):       # whatever is still below the launch threshold
            self._done.discard("__flush__"); self.stages["__flush__"] = []
            self.stage_done("__flush__", flush=True)
            del self.stages["__flush__"]
        for works, t, w in self._pending:
            for work in works:
                work.wait()
            if w is not None:
                from fourm.hip import ops
                (self._unpack or ops.bf16_to_f32_scaled)(w, t, 1.0 / self.world)          # unpack + mean into the fp32 gradient store
            elif not self._avg:
                t.div_(self.world)
        self._pending = []
        if timed:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self._exch_events.append((e0, e1))


# Compute units the persistent GEMM grids leave to the collectives in "overlap" mode: 16 = two per XCD.  Measured on one MI355X
# (profiles/r04_reserved_cus.txt): 61.4 ms per 4M-B step on 256 CUs, 62.5 on 240, 62.7 on 248 (the tile-width choice of the NT GEMM and the
# cut of the weight-gradient lists follow the grid).  ``cap_collective_channels`` bounds RCCL to as many channels (= resident workgroups).
DEFAULT_RESERVED_CUS = 16


def cap_collective_channels(n: Optional[int] = None):
    """Call BEFORE the process group's first collective (RCCL reads it when the communicator is created): at most ``n`` channels, so that a
    collective's workgroups fit the CUs DataParallel reserves.  4M-B needs ~2.5 GB over each GPU's links per ~45 ms of backward = 56 GB/s:
    a fraction of what 16 channels move.  An explicit NCCL_MAX_NCHANNELS in the environment wins; FOURM_DP_CAP_CHANNELS=0 switches the cap
    off (it is process-wide: every collective of the process group, the tokenizers' codebook all-reduce included, runs on <= n channels);
    FOURM_DP_EXCHANGE=tail never sets it.  The value is NOT validated on N > 1 hardware: it is printed when it is set, and
    bench.py records it in ``config.data_parallel``.  Returns the cap it set, else None."""
    import os
    if n is None:
        n = int(os.environ.get("FOURM_DP_RESERVED_CUS", str(DEFAULT_RESERVED_CUS)))
    if os.environ.get("FOURM_DP_CAP_CHANNELS", "1") == "0" or "NCCL_MAX_NCHANNELS" in os.environ:
        return None
    if n > 0 and os.environ.get("FOURM_DP_EXCHANGE", "overlap").lower() == "overlap":
        os.environ["NCCL_MAX_NCHANNELS"] = str(n)
        if int(os.environ.get("RANK", "0")) == 0:
            import sys
            print(f"[fourm.parallel] NCCL_MAX_NCHANNELS={n} (collectives confined to the {n} CUs the GEMM grids leave free; "
                  f"FOURM_DP_CAP_CHANNELS=0 or FOURM_DP_EXCHANGE=tail to lift it)", file=sys.stderr, flush=True)      # stderr: bench.py's stdout is ONE JSON line
        return n
    return None


class DataParallel(nn.Module):
    """Drop-in for the way the trainer uses DDP: callable like the module, ``.module``, ``.no_sync()``,
    ``.parameters()``, ``.train()/.eval()`` (run_training_4m.py:512-513, :721, :736).

    Replicas start from rank 0's weights (broadcast at construction, like DDP).  Inside ``no_sync()``
    gradients only accumulate locally; the first backward outside it exchanges the accumulated mean."""

    def __init__(self, module: nn.Module, device_ids=None, find_unused_parameters: bool = False, process_group=None,
                 bucket_mb: int = 256, algorithm: str = "all_reduce", wire_dtype=None, reserved_cus: Optional[int] = None,
                 force_collectives: bool = False, min_launch_mb: Optional[float] = None, exchange: Optional[str] = None, comm: Optional[str] = None):
        """``exchange`` (env FOURM_DP_EXCHANGE): "overlap" (default) - a stage's gradient slices are exchanged while the next stage computes;
        "tail" - nothing is exchanged under the backward, the whole gradient store goes out in bucket-sized collectives when the backward
        has finished (4M-B: 1.44 GB fp32; a ring all-reduce moves 2 (N - 1) / N of it over each GPU's links).

        ``reserved_cus`` (env FOURM_DP_RESERVED_CUS; only meaningful with "overlap"): compute units the persistent GEMM grids leave free
        while gradients are exchanged.  What can share a CU is a question of registers and LDS (profiles/r04_kernel_resources.txt,
        tools/kernel_resources.py): a workgroup of the dense NT GEMM (gemm_nt3: 201 - 256 VGPRs x 2 waves per SIMD, 160 KB of LDS with its
        staged epilogue) or of the weight-gradient GEMM (gemm_tn_multi: 256 VGPRs x 2, 96 KB) owns its CU outright - nothing of another
        stream can be co-resident with it (round 3's docstring said otherwise; it was wrong).  A collective's workgroups are long-lived
        (one per channel for the whole message), so while a collective is resident every persistent GEMM grid of 256 workgroups finds some
        CUs taken and runs those workgroups in a second round: reserved_cus >= the collective's channel count (bound it with
        NCCL_MAX_NCHANNELS) keeps the grids and the collective on disjoint CUs.  The price on one GPU is what the tilings lose on 248 CUs
        (profiles/r04_reserved_cus.txt); the streaming kernels (LayerNorm, activation backward, AdamW: <= 152 VGPRs, no LDS) share CUs
        freely.  No N > 1 hardware was available to choose between "overlap" + reservation and "tail": both are one environment variable."""
        super().__init__()
        import os
        self.module = module
        self.process_group = process_group
        self._sync = True
        self._bucket_elems = bucket_mb * 1024 * 1024 // 4
        self._algorithm, self._wire, self._force = algorithm, wire_dtype, force_collectives
        self._exchange_mode = (exchange or os.environ.get("FOURM_DP_EXCHANGE", "overlap")).lower()
        if self._exchange_mode not in ("overlap", "tail"):
            raise ValueError(f"exchange {self._exchange_mode!r}: 'overlap' or 'tail'")
        # launch a collective only when this much gradient is waiting (env FOURM_DP_MIN_LAUNCH_MB; default 192 MB: ~8 per 4M-B step)
        self._min_launch_mb = float(os.environ.get("FOURM_DP_MIN_LAUNCH_MB", "192")) if min_launch_mb is None else float(min_launch_mb)
        if self._exchange_mode == "tail":
            self._min_launch_mb = float("inf")           # everything waits for GradReducer.finish()
        if reserved_cus is None:
            reserved_cus = int(os.environ.get("FOURM_DP_RESERVED_CUS", str(DEFAULT_RESERVED_CUS)))
        self._reserved_cus = 0 if self._exchange_mode == "tail" else reserved_cus
        self._comm = (comm or os.environ.get("FOURM_DP_COMM", "torch")).lower()      # "direct": fourm.parallel.rccl (ctypes RCCL on a side stream)
        self._reducer: Optional[GradReducer] = None
        self._reducer_for = None
        if dist.is_initialized() and (dist.get_world_size(process_group) > 1 or force_collectives):
            self.broadcast_from_rank0()

    def broadcast_from_rank0(self):
        """Replicas start from rank 0's weights (what DDP does at construction, run_training_4m.py:512) - as TWO collectives instead
        of one per tensor (~360 for 4M-B): the engine's flat fp32 parameter store in one piece (large message: what reaches xGMI
        link bandwidth), and every buffer / stray parameter packed into one staging tensor per dtype."""
        module, group = self.module, self.process_group
        from fourm.hip.engine import bump_weight_epoch
        with torch.no_grad():
            done = set()
            eng = getattr(module, "engine", None) if any(p.is_cuda for p in module.parameters()) else None
            if eng is not None:
                eng._ensure_flat()
                dist.broadcast(eng.flat_params, src=0, group=group)
                lo, hi = eng.flat_params.data_ptr(), eng.flat_params.data_ptr() + eng.flat_params.numel() * 4
                done = {id(p) for p in module.parameters() if lo <= p.data_ptr() < hi}
            rest = [t for t in list(module.parameters()) + list(module.buffers()) if id(t) not in done]
            by_type = {}
            for t in rest:
                by_type.setdefault((t.dtype, t.device), []).append(t)
            for ts in by_type.values():
                packed = torch.cat([t.detach().reshape(-1) for t in ts])
                dist.broadcast(packed, src=0, group=group)
                o = 0
                for t in ts:
                    t.data.copy_(packed[o:o + t.numel()].view_as(t))
                    o += t.numel()
            self.n_init_broadcasts = (1 if eng is not None else 0) + len(by_type)
        bump_weight_epoch()

    @contextlib.contextmanager
    def no_sync(self):
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _attach(self):
        eng = self.module.engine
        eng._ensure_flat()
        if self._reducer is None or self._reducer_for is not eng.flat_grads:
            self._reducer = GradReducer(eng.flat_grads, eng.grad_stages(), self.process_group, self._bucket_elems,
                                        algorithm=self._algorithm, wire_dtype=self._wire, force=self._force, min_launch_mb=self._min_launch_mb,
                                        comm=self._comm if eng.flat_grads.is_cuda and dist.get_backend(self.process_group) == "nccl" else "torch")
            self._reducer_for = eng.flat_grads
            self._reducer.time_exchange = bool(getattr(self, "time_exchange", False))
            # RCCL's kernels run beside the persistent GEMM grids: the engine reserves the CUs for the span of the backward (begin() ... finish())
            self._reducer.reserved_cus = self._reserved_cus if (eng.flat_grads.is_cuda and dist.get_backend(self.process_group) == "nccl") else 0
        eng.reducer = self._reducer if self._sync else None

    def forward(self, *args, **kwargs):
        if torch.is_grad_enabled() and dist.is_initialized() and (dist.get_world_size(self.process_group) > 1 or self._force):
            self._attach()
        elif hasattr(self.module, "_engine") and self.module._engine is not None:
            self.module._engine.reducer = None
        return self.module(*args, **kwargs)
