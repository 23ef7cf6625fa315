"""Host-side cost of one train step: time to ENQUEUE a step (no device sync) vs time to finish it, plus the
number of shadow-table rebuilds / refresh launches.  python tools/host_overhead.py [--batch 256]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=256); ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--cprofile", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
from fourm.data.synthetic import synthetic_batch
from fourm.hip import ops, engine as E
from fourm.utils.optim_factory import FusedAdamW, get_parameter_groups
import contextlib, io
model = bench.build_model("fm_base_12e_12d_swiglu_nobias", dev).train()
with contextlib.redirect_stdout(io.StringIO()):
    groups = get_parameter_groups(model, weight_decay=0.05, skip_list=model.no_weight_decay())
opt = FusedAdamW(groups, lr=1e-4, betas=(0.9, 0.95), eps=1e-8)
batches = [synthetic_batch(model, a.batch, 128, 128, device=dev, seed=i) for i in range(2)]
counts = {"table": 0, "refresh": 0}
_t, _r = ops.shadow_jobs_table, ops.shadow_refresh
def t2(*x, **k): counts["table"] += 1; return _t(*x, **k)
def r2(*x, **k): counts["refresh"] += 1; return _r(*x, **k)
ops.shadow_jobs_table, ops.shadow_refresh = t2, r2

def step(i):
    t0 = time.perf_counter()
    loss, _ = model(batches[i % 2], 128, 128, loss_type="mod")
    t1 = time.perf_counter()
    loss.backward()
    t2_ = time.perf_counter()
    opt.fused_grad_norm(); opt.step(); opt.zero_grad(set_to_none=True)
    t3 = time.perf_counter()
    return t1 - t0, t2_ - t1, t3 - t2_

for i in range(2):
    step(i)
torch.cuda.synchronize()
print("after warmup", counts)
if a.cprofile:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
for i in range(a.steps):
    c0 = dict(counts)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f, b, o = step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2_ = time.perf_counter()
    print(f"step {i}: enqueue fwd {f*1e3:.1f} bwd {b*1e3:.1f} opt {o*1e3:.1f} = {(t1-t0)*1e3:.1f} ms; finished after {(t2_-t0)*1e3:.1f} ms; "
          f"tables +{counts['table']-c0['table']} refresh +{counts['refresh']-c0['refresh']}")
if a.cprofile:
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)

# ---- the same step replayed from one hipGraph (fourm/hip/graph.py) ------------------------------------------------------------
from fourm.hip.graph import GraphedTrainStep
gs = GraphedTrainStep(model, opt, batches[0], 128, 128, warmup=1)
for i in range(2):
    gs.step(batches[i % 2])
torch.cuda.synchronize()
for i in range(a.steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    gs.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2_ = time.perf_counter()
    print(f"graph step {i}: enqueue {(t1-t0)*1e3:.2f} ms (static batch); finished after {(t2_-t0)*1e3:.1f} ms")
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(10):
    gs.step(batches[i % 2])
torch.cuda.synchronize()
print(f"graph: 10 steps with batch copies {(time.perf_counter()-t0)*100:.2f} ms/step; final loss {float(gs.loss):.4f}")
