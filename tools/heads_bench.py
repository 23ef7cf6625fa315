#!/usr/bin/env python3
"""Micro-benchmark of the per-modality head GEMMs (grouped NT fwd / dX, grouped TN dW) and cross-entropy at the
4M-B mod7 shapes: 32768 decoder rows split over 7 vocabularies.  python tools/heads_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch
from fourm.hip import ops, _lib as L

dev = "cuda"
D, R = 768, 256 * 128
vocabs = [int(v) for v in os.environ.get("VOCABS", "16384,8192,8192,4096,8192,30000,30000").split(",")]
nH = len(vocabs)
g = torch.Generator().manual_seed(0)
head = torch.randint(0, nH, (R,), generator=g).int().to(dev)
Rp = ops.padded_rows(R, nH)
i32 = torch.int32
seg_start, seg_count = torch.zeros(nH, dtype=i32, device=dev), torch.zeros(nH, dtype=i32, device=dev)
perm, r2p = torch.zeros(Rp, dtype=i32, device=dev), torch.zeros(R, dtype=i32, device=dev)
tile_group = torch.zeros(Rp // ops.SEG, dtype=i32, device=dev)
ops.segment_rows(head, nH, seg_start, seg_count, perm, r2p, tile_group)
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.3).to(torch.bfloat16)
yp = rnd(Rp, D)
ws = [rnd(v, D) for v in vocabs]
wts = [torch.zeros(D, ops.ru(v, 64), device=dev, dtype=torch.bfloat16) for v in vocabs]
for w, t in zip(ws, wts):
    t[:, :w.shape[0]] = w.t()
ldl = ops.ru(max(vocabs), 64)
logits = torch.zeros(Rp, ldl, device=dev, dtype=torch.bfloat16)
g_fwd = ops.make_groups([dict(W=w, N=v, K=D, ldw=D) for w, v in zip(ws, vocabs)], dev)
g_bwd = ops.make_groups([dict(W=t, N=D, K=ops.ru(v, 64), ldw=ops.ru(v, 64)) for t, v in zip(wts, vocabs)], dev)
dws = [torch.zeros(v, D, device=dev) for v in vocabs]
g_tn = ops.make_groups([dict(out=o, N=v) for o, v in zip(dws, vocabs)], dev)
dyp = torch.zeros(Rp, D, device=dev, dtype=torch.bfloat16)
counts = seg_count.tolist()
flops = 2.0 * D * sum(c * v for c, v in zip(counts, vocabs))


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def rep(name, t, fl=flops):
    print(f"{name:44s} {t * 1e6:9.1f} us {fl / t / 1e12:8.1f} TF/s", flush=True)


rep("grouped NT logits (fwd)", timeit(lambda: ops.gemm_nt_grouped(yp, g_fwd, tile_group, logits, max(vocabs), max_K=D)))
rep("grouped NT dY (bwd)", timeit(lambda: ops.gemm_nt_grouped(logits, g_bwd, tile_group, dyp, D, max_K=ldl)))
rep("grouped TN dW heads", timeit(lambda: ops.gemm_tn_grouped(logits, yp, g_tn, seg_start, seg_count, nH, max(vocabs), Rp, D)))
rep("dense per head, device-side rows (fwd)", timeit(lambda: ops.gemm_nt_heads(yp, ws, vocabs, seg_start, seg_count, logits, D)))
# the same work as 7 dense launches (upper bound on what the grouped kernel could do)
def dense_default():
    for h in range(nH):
        c = counts[h]
        ops.gemm_nt(yp[:ops.ru(c, 256)], ws[h], logits[:ops.ru(c, 256)], N=vocabs[h], K=D, M=c)
rep("7 dense NT launches, default dispatch (fwd)", timeit(dense_default))
for cfg in (2, 6):
    L.lib.fm_set_gemm_nt_config(cfg)
    def dense():
        for h in range(nH):
            s, c = int(seg_start[h]) if False else 0, counts[h]
            ops.gemm_nt(yp[:ops.ru(c, 128)], ws[h], logits[:ops.ru(c, 128)], N=vocabs[h], K=D, M=c)
    rep(f"7 dense NT launches cfg{cfg} (fwd)", timeit(dense))
L.lib.fm_set_gemm_nt_config(9 + 256)
tgt = torch.randint(0, 4096, (R,), device=dev)
vt = torch.tensor(vocabs, dtype=i32, device=dev)
row_loss, row_lse = torch.zeros(Rp, device=dev), torch.zeros(Rp, device=dev)
head_loss, total = torch.zeros(nH, device=dev), torch.zeros(1, device=dev)
nb = 2.0 * sum(c * v for c, v in zip(counts, vocabs))
t = timeit(lambda: ops.cross_entropy(logits, perm, tile_group, tgt, vt, seg_start, seg_count, nH, max(vocabs), row_loss, row_lse, head_loss, total))
print(f"{'cross-entropy fwd':44s} {t * 1e6:9.1f} us {nb / t / 1e9:8.1f} GB/s")
gs = torch.ones(1, device=dev)
t = timeit(lambda: ops.cross_entropy(logits, perm, tile_group, tgt, vt, seg_start, seg_count, nH, max(vocabs), row_loss, row_lse, head_loss, total, grad_scale=gs, write_grad=True))
print(f"{'cross-entropy bwd (in place)':44s} {t * 1e6:9.1f} us {2 * nb / t / 1e9:8.1f} GB/s")
