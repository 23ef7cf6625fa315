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
    """All-reduce (mean) of slices of a flat gradient buffer, launched stage by stage.

    ranges_by_stage: {stage: [(offset, length), ...]} in elements of ``flat``.  Adjacent ranges are
    merged; ranges larger than ``bucket_elems`` are split so that several collectives are in flight."""

    def __init__(self, flat: torch.Tensor, ranges_by_stage: Dict[str, Sequence[Tuple[int, int]]], group=None,
                 bucket_elems: int = 64 * 1024 * 1024):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = bucket_elems
        self.stages = {s: self._merge(r) for s, r in ranges_by_stage.items()}
        covered = sorted((o, n) for r in self.stages.values() for o, n in r)
        for (o1, n1), (o2, _) in zip(covered, covered[1:]):
            if o1 + n1 > o2:
                raise ValueError("gradient ranges of different stages overlap")
        self._pending = []
        self._done = set()
        backend = dist.get_backend(group) if dist.is_initialized() else ""
        self._avg = backend == "nccl"          # RCCL averages in the collective; gloo has no AVG

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
        self._pending, self._done = [], set()

    def stage_done(self, stage: str):
        """Gradients of ``stage`` are final on the compute stream: start their exchange."""
        if self.world == 1 or stage in self._done or stage not in self.stages:
            return
        self._done.add(stage)
        for o, n in self.stages[stage]:
            t = self.flat[o:o + n]
            op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
            self._pending.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), t))

    def finish(self):
        """Launch whatever stage has not been reported, then wait for every collective."""
        for s in self.stages:
            self.stage_done(s)
        for work, t in self._pending:
            work.wait()
            if not self._avg:
                t.div_(self.world)
        self._pending = []


class DataParallel(nn.Module):
    """Drop-in for the way the trainer uses DDP: callable like the module, ``.module``, ``.no_sync()``,
    ``.parameters()``, ``.train()/.eval()`` (run_training_4m.py:512-513, :721, :736).

    Replicas start from rank 0's weights (broadcast at construction, like DDP).  Inside ``no_sync()``
    gradients only accumulate locally; the first backward outside it exchanges the accumulated mean."""

    def __init__(self, module: nn.Module, device_ids=None, find_unused_parameters: bool = False, process_group=None,
                 bucket_mb: int = 256):
        super().__init__()
        self.module = module
        self.process_group = process_group
        self._sync = True
        self._bucket_elems = bucket_mb * 1024 * 1024 // 4
        self._reducer: Optional[GradReducer] = None
        self._reducer_for = None
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
            from fourm.hip.engine import bump_weight_epoch
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
            self._reducer = GradReducer(eng.flat_grads, eng.grad_stages(), self.process_group, self._bucket_elems)
            self._reducer_for = eng.flat_grads
        eng.reducer = self._reducer if self._sync else None

    def forward(self, *args, **kwargs):
        if torch.is_grad_enabled() and dist.is_initialized() and dist.get_world_size(self.process_group) > 1:
            self._attach()
        elif hasattr(self.module, "_engine") and self.module._engine is not None:
            self.module._engine.reducer = None
        return self.module(*args, **kwargs)
