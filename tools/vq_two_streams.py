#!/usr/bin/env python3
"""RGB tokenizer throughput with ONE and with TWO sub-batches of 64 in flight (two model replicas = two workspaces, two HIP streams): how much of
the tile-quantisation loss of the 12 544-row GEMMs (147 - 196 tiles on 256 CUs) a second stream's kernels fill."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.vq import VQ  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
m0 = VQ(image_size=224, enc_type="vit_b_enc", patch_size=16, post_mlp=True, codebook_size=16384, latent_dim=32, norm_codes=True, sync_codebook=False).to(dev).eval()
m1 = copy.deepcopy(m0)
xs = [torch.rand(64, 3, 224, 224, device=dev) * 2 - 1 for _ in range(2)]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
for m, x in ((m0, xs[0]), (m1, xs[1])):
    for _ in range(3):
        m.tokenize(x)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for i in range(n):
    m0.tokenize(xs[i % 2])
torch.cuda.synchronize()
one = 64 * n / (time.perf_counter() - t0)
for j, (m, x) in enumerate(((m0, xs[0]), (m1, xs[1]))):
    with torch.cuda.stream(s[j]):
        for _ in range(2):
            m.tokenize(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    j = i % 2
    with torch.cuda.stream(s[j]):
        (m0 if j == 0 else m1).tokenize(xs[j])
torch.cuda.synchronize()
two = 64 * n / (time.perf_counter() - t0)
print(f"one sub-batch in flight: {one:8.0f} images/s;  two (two streams, two workspaces): {two:8.0f} images/s  ({two / one:.3f} x)")
