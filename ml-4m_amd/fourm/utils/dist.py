"""Process-group bring-up: one process per GPU, ``torch.distributed`` with the ``nccl`` backend (RCCL on
ROCm, collectives over xGMI).  Reads the torchrun environment like upstream ``fourm/utils/dist.py:78-99``;
falls back to ``gloo`` when no GPU is visible (CPU tests of the data-parallel logic)."""
import datetime
import os

import torch
import torch.distributed as dist


def is_dist_avail_and_initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size() -> int:
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank() -> int:
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process() -> bool:
    return get_rank() == 0


def init_distributed_mode(args):
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        args.rank, args.world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    else:
        args.rank, args.world_size, args.gpu, args.distributed = 0, 1, 0, False
        return
    args.distributed = True
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(args.gpu)
    args.dist_backend = "nccl" if use_gpu else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if use_gpu:
        from fourm.parallel import cap_collective_channels
        cap_collective_channels()             # RCCL's workgroups fit the CUs fourm.parallel.DataParallel keeps free of GEMM grids
    dist.init_process_group(backend=args.dist_backend, init_method=getattr(args, "dist_url", "env://"),
                            world_size=args.world_size, rank=args.rank, timeout=datetime.timedelta(minutes=80))
    dist.barrier()
    setup_for_distributed(args.rank == 0 or bool(getattr(args, "print_all", False)))


def setup_for_distributed(is_master: bool):
    """print() is silent on every rank but the master (or everywhere with --print_all), ``force=True`` overrides
    (upstream fourm/utils/dist.py setup_for_distributed: 8 ranks otherwise print args, model and every meter line 8 times)."""
    import builtins
    builtin_print = getattr(builtins.print, "_fourm_builtin", builtins.print)

    def rank_print(*a, **kw):
        if kw.pop("force", False) or is_master:
            builtin_print(*a, **kw)
    rank_print._fourm_builtin = builtin_print
    builtins.print = rank_print


# names only upstream's same-named module defines (see fourm/_upstream.py)
from fourm import _upstream as _up
_up.merge(__name__, globals())
