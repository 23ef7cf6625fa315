"""Pin of oracle/vq_oracle.codebook_ema_update against the UNMODIFIED upstream CosineSimCodebook in training mode (build container
only: needs /root/reference).  CPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CHILD = r'''
import sys, os
sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, ROOT)
import ref_stubs; ref_stubs.install()
sys.path.insert(0, "/root/reference")
import torch
from fourm.vq.quantizers.quantize_lucid import CosineSimCodebook
from oracle import vq_oracle as V
torch.manual_seed(0)
for K, R, thr in ((64, 500, 0.0), (256, 300, 0.25), (128, 40, 2.0)):
    cb = CosineSimCodebook(dim=32, codebook_size=K, decay=0.9, threshold_ema_dead_code=thr, code_replacement_policy="batch_random")
    cb.cluster_size.copy_(torch.rand(K) * 3)
    cb.train()
    z = torch.randn(2, R // 2, 32) * 1.7
    e0, c0 = cb.embed.clone(), cb.cluster_size.clone()
    torch.manual_seed(123)
    quant, ind = cb(z)
    ind_o, _ = V.assign_codes(z.reshape(-1, 32), e0)
    assert torch.equal(ind.reshape(-1), ind_o)
    # the oracle's dead-code rows: what sample_vectors draws with the same seed (randperm is the first RNG use of the forward)
    e1, c1 = V.codebook_ema_update(e0, c0, z, ind, 0.9)
    rows = None
    if thr > 0:
        dead = c1 < thr
        n = int(dead.sum())
        if n:
            torch.manual_seed(123)
            Rr = z.reshape(-1, 32).shape[0]
            idx = torch.randperm(Rr)[:n] if Rr >= n else torch.randint(0, Rr, (n,))
            rows = torch.nn.functional.normalize(z.reshape(-1, 32)[idx], dim=-1)
    e1, c1 = V.codebook_ema_update(e0, c0, z, ind, 0.9, threshold_dead=thr, replace_rows=rows)
    assert torch.allclose(c1, cb.cluster_size, rtol=1e-6, atol=1e-7), (K, (c1 - cb.cluster_size).abs().max())
    assert torch.allclose(e1, cb.embed, rtol=1e-5, atol=1e-7), (K, (e1 - cb.embed).abs().max())
    print("ok", K, R, thr, int((c1 < thr).sum()) if thr else 0)
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not on this machine")
def test_codebook_ema_update_matches_upstream():
    # (a child process: `fourm` must resolve to the reference tree there, not to this package)
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert p.stdout.count("ok") == 3
