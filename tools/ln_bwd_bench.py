#!/usr/bin/env python3
"""LayerNorm backward at the 4M-B row count (32768 x 768): x_hat from the fp32 input (fm_layernorm_bwd) against x_hat from the saved bf16
norm output (fm_layernorm_bwd_h); us per launch and bytes moved per second."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402


def main():
    from fourm.hip import ops
    dev = "cuda"
    R, D = 32768, 768
    torch.manual_seed(0)
    NB = 6            # rotate over buffer sets larger than the 256 MB cache
    sets = []
    for _ in range(NB):
        x = torch.randn(R, D, device=dev)
        w = torch.rand(D, device=dev) + 0.5
        h = torch.zeros(R, D, device=dev, dtype=torch.bfloat16)
        mean, rstd = torch.zeros(R, device=dev), torch.zeros(R, device=dev)
        ops.layernorm_fwd(x, w, None, h, mean, rstd)
        sets.append(dict(x=x, w=w, h=h, mean=mean, rstd=rstd, dy=torch.randn(R, D, device=dev).bfloat16(), g=torch.randn(R, D, device=dev),
                         gb=torch.zeros(R, D, device=dev, dtype=torch.bfloat16), dw=torch.zeros(D, device=dev)))
    for name, use_h, bpe, with_dw in (("from x (16 B/elem)", False, 16, True), ("from h (14 B/elem)", True, 14, True),
                                      ("from x, no dw", False, 16, False), ("from h, no dw", True, 14, False)):
        def run(s):
            ops.layernorm_bwd(s["dy"], s["x"], s["w"], s["mean"], s["rstd"], s["g"], dres=s["g"], dx_bf16=s["gb"], dw=s["dw"] if with_dw else None,
                              h=s["h"] if use_h else None)
        for s in sets:
            run(s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            for s in sets:
                run(s)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (n * NB)
        print(f"{name}: {us:7.1f} us per launch  {R * D * bpe / us / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
