#!/usr/bin/env python3
"""What a 1-GPU box can say about the gradient exchange (VERDICT r02 #5b): RCCL at world size 1 with the collectives FORCED
(``DataParallel(force_collectives=True)``), 4M-B mod7 at batch 256.
  * step time without / with the exchange (all-reduce fp32, reduce-scatter + all-gather, bf16 wire), interleaved;
  * under ``rocprofv3 --kernel-trace`` (tools/overlap_dp.sh) the trace of the same run gives, per step, when the first / last RCCL
    kernel ran relative to the backward's first / last GEMM and how much of the RCCL kernel time lies under compute kernels.
At world size 1 a collective moves no bytes over xGMI: this measures scheduling (do RCCL's kernels get CUs while the persistent GEMM
grids run?) and host-side cost, not link bandwidth."""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-4m_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--modes", default="none,all_reduce,reduce_scatter,all_reduce_bf16")
    ap.add_argument("--reserved-cus", type=int, default=0)
    ap.add_argument("--min-launch-mb", type=float, default=None)
    a = ap.parse_args()
    import bench
    from fourm.data.synthetic import synthetic_batch
    from fourm.parallel import DataParallel
    from fourm.utils.optim_factory import FusedAdamW, get_parameter_groups
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    torch.manual_seed(0)
    model = bench.build_model("fm_base_12e_12d_swiglu_nobias", dev, "mod7").train()
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        groups = get_parameter_groups(model, weight_decay=0.05, skip_list=model.no_weight_decay())
    opt = FusedAdamW(groups, lr=1e-4, betas=(0.9, 0.95), eps=1e-8)
    batches = [synthetic_batch(model, a.batch, 128, 128, device=dev, seed=i) for i in range(2)]
    wrappers = {"none": None}
    for m in a.modes.split(","):
        if m == "none":
            continue
        kw = dict(algorithm="reduce_scatter") if m.startswith("reduce_scatter") else {}
        if m.endswith("bf16"):
            kw["wire_dtype"] = torch.bfloat16
        wrappers[m] = DataParallel(model, force_collectives=True, reserved_cus=a.reserved_cus, min_launch_mb=a.min_launch_mb, **kw)

    def step(fwd, i):
        random.seed(0)
        loss, _ = fwd(batches[i % 2], 128, 128, loss_type="mod")
        loss.backward()
        opt.fused_grad_norm(); opt.step(); opt.zero_grad(set_to_none=True)

    res = {m: [] for m in wrappers}
    for rep in range(3):
        for m, w in wrappers.items():
            fwd = w if w is not None else model
            if w is None:
                model.engine.reducer = None
            for i in range(2):
                step(fwd, i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                step(fwd, i)
            torch.cuda.synchronize()
            res[m].append(1e3 * (time.perf_counter() - t0) / a.steps)
    out = {m: min(v) for m, v in res.items()}
    print(json.dumps({"ms_per_step": out, "reserved_cus": a.reserved_cus, "batch": a.batch,
                      "stages": len(wrappers[next(k for k in wrappers if k != 'none')]._reducer.stages) if len(wrappers) > 1 else 0}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
