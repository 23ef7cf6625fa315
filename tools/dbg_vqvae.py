import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import numpy as np, torch, torch.nn.functional as F
from tests.test_vqvae import case, build
name = sys.argv[1] if len(sys.argv) > 1 else "vqvae_small"
c, cfg, sd, x, g = case(name)
m = build(c, cfg); m.load_state_dict(sd); m = m.cuda().train()
xg = x.cuda()
dec, cl = m(xg)
(F.mse_loss(dec, xg) + cl.sum()).backward()
torch.cuda.synchronize()
for k, p in m.named_parameters():
    if p.grad is None: continue
    ref = float(g["grad_l2/" + k]) if "grad_l2/" + k in g.files else float("nan")
    n = float(p.grad.double().norm())
    flag = "NAN" if not np.isfinite(n) else ""
    if flag or abs(n - ref) / (ref + 1e-12) > 0.05 or k.endswith(("out_proj.weight", "quant_proj.weight", "proj.weight")):
        print(f"{k:50s} got {n:12.5e} ref {ref:12.5e} {flag}")
