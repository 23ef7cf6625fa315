"""RCCL called directly (ctypes on torch's own librccl.so), outside ``torch.distributed``'s process group.

Why: a collective issued through torch's ProcessGroupNCCL is watched by torch's watchdog thread, which polls the collective's completion
events - inside a hipGraph capture that is ``hipErrorCapturedEvent`` and the process dies (round 4).  ``ncclAllReduce`` on a
plain ``hipStream_t`` has no such observer: the gradient exchange becomes a node of the captured train step like any kernel, and the
eager path loses torch's per-collective host overhead (~0.1 ms each).

The communicator is created from a ``ncclUniqueId`` made on rank 0 and handed to the other ranks through ``torch.distributed`` (any
backend: it is 128 bytes, once).  Only what the data-parallel exchange needs is bound: all-reduce (sum / avg; f32 / bf16), reduce-scatter,
all-gather.  STATUS: exercised on one GPU (world size 1 and the gloo-bootstrapped two-process test); no N > 1 RCCL run exists yet.
Select with ``DataParallel(comm="direct")`` or FOURM_DP_COMM=direct; the default stays torch.distributed."""
import ctypes as C
import os
from typing import Optional

import torch
import torch.distributed as dist

NCCL_SUM, NCCL_AVG = 0, 4
NCCL_FLOAT32, NCCL_BFLOAT16 = 7, 9


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]


_lib = None


def _load():
    global _lib
    if _lib is None:
        cands = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so", "librccl.so"]
        err = None
        for c in cands:
            try:
                _lib = C.CDLL(c)
                break
            except OSError as e:      # noqa: PERF203
                err = e
        if _lib is None:
            raise ImportError(f"librccl.so not found ({err})")
        _lib.ncclGetErrorString.restype = C.c_char_p
        _lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
        _lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _UniqueId, C.c_int]
        _lib.ncclCommDestroy.argtypes = [C.c_void_p]
        _lib.ncclAllReduce.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.ncclReduceScatter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _lib.ncclAllGather.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"RCCL {what}: {_load().ncclGetErrorString(rc).decode()} (rc={rc})")


def _dtype(t):
    if t.dtype == torch.float32:
        return NCCL_FLOAT32
    if t.dtype == torch.bfloat16:
        return NCCL_BFLOAT16
    raise TypeError(f"direct RCCL exchange: dtype {t.dtype}")


class DirectComm:
    """One RCCL communicator over the ranks of ``group`` (default: the world), bound to this process's current device."""

    def __init__(self, group=None, device: Optional[torch.device] = None):
        lib = _load()
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        uid = _UniqueId()
        if self.rank == 0:
            _check(lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            raw = [C.string_at(C.addressof(uid), 128) if self.rank == 0 else None]      # (string_at: all 128 bytes, NULs included)
            dist.broadcast_object_list(raw, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            C.memmove(C.addressof(uid), raw[0], 128)
        self._comm = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.ncclCommInitRank(C.byref(self._comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.stream = torch.cuda.Stream(device=self.device)            # the exchange's own stream (fenced with events against the compute stream)

    def all_reduce_(self, t: torch.Tensor, avg: bool, stream: Optional[torch.cuda.Stream] = None):
        s = (stream or self.stream).cuda_stream
        _check(_load().ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), _dtype(t), NCCL_AVG if avg else NCCL_SUM, self._comm, s), "ncclAllReduce")

    def reduce_scatter(self, shard: torch.Tensor, t: torch.Tensor, avg: bool, stream=None):
        s = (stream or self.stream).cuda_stream
        _check(_load().ncclReduceScatter(t.data_ptr(), shard.data_ptr(), shard.numel(), _dtype(t), NCCL_AVG if avg else NCCL_SUM, self._comm, s), "ncclReduceScatter")

    def all_gather(self, t: torch.Tensor, shard: torch.Tensor, stream=None):
        s = (stream or self.stream).cuda_stream
        _check(_load().ncclAllGather(shard.data_ptr(), t.data_ptr(), shard.numel(), _dtype(t), self._comm, s), "ncclAllGather")

    def close(self):
        if self._comm:
            _load().ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()


_COMMS = {}


def shared_comm(group=None, device: Optional[torch.device] = None) -> DirectComm:
    """ONE communicator per (process group, device), created on first use and reused by every gradient reducer built afterwards (the data-parallel
    wrapper rebuilds its reducer whenever the flat gradient store is re-allocated: a communicator per rebuild would leak - ncclCommInitRank is a
    collective and nothing destroyed the old ones).  Destroyed by close_all() / at interpreter exit."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    key = (id(group) if group is not None else None, device.index)
    c = _COMMS.get(key)
    if c is None or not c._comm:
        c = _COMMS[key] = DirectComm(group, device)
    return c


def close_all():
    for c in list(_COMMS.values()):
        c.close()
    _COMMS.clear()


import atexit  # noqa: E402
atexit.register(close_all)


class _DirectWork:
    """Stand-in for torch's Work handle: wait() = the CURRENT stream waits for what the exchange stream has been given so far."""

    def __init__(self, comm: DirectComm):
        self.ev = torch.cuda.Event()
        self.ev.record(comm.stream)

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)
