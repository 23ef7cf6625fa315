#!/usr/bin/env python3
"""What the CU reservation of the data-parallel mode costs on ONE GPU (world size 1, collectives forced): 4M-B mod7 train step, batch 256, with
fourm.parallel.DataParallel(reserved_cus=16, force_collectives=True).  FOURM_DP_RESERVE_ALWAYS=1: reserved for the whole step (rounds 4 - 5);
default: from the backward's begin() to its finish()."""
import os, sys, time, datetime
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=180))
from bench import build_model
from fourm.data.synthetic import synthetic_batch
from fourm.parallel import DataParallel
from fourm.utils.optim_factory import FusedAdamW
dev = "cuda"
model = build_model("fm_base_12e_12d_swiglu_nobias", dev, "mod7").train()
opt = FusedAdamW([{"params": [p for p in model.parameters() if p.dim() > 1], "weight_decay": 0.05}, {"params": [p for p in model.parameters() if p.dim() <= 1], "weight_decay": 0.0}], lr=1e-4, betas=(0.9, 0.95))
res = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dp = DataParallel(model, force_collectives=True, reserved_cus=res)
batches = [synthetic_batch(model, 256, 128, 128, device=dev, seed=i) for i in range(2)]


def step(i):
    loss, _ = dp(batches[i % 2], 128, 128)
    loss.backward(); opt.fused_grad_norm(clip=None, lazy=True); opt.step(); opt.zero_grad(set_to_none=True)


for i in range(3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for i in range(n):
    step(i)
torch.cuda.synchronize()
print(f"reserved {res} CUs, FOURM_DP_RESERVE_ALWAYS={os.environ.get('FOURM_DP_RESERVE_ALWAYS', '0')}: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per step (world size 1, forced collectives)")
dist.destroy_process_group()
