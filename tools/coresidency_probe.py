#!/usr/bin/env python3
"""Can an HBM-bound kernel run UNDER a weight-gradient GEMM on the same CUs?  

The dW launch of a layer (fm_gemm_tn_multi) is off the dX critical path.  A workgroup of the default dW kernel owns its CU
(512 threads x 256 registers = the whole register file): nothing of another stream can be co-resident.  The 128 x 256-tile form
(fm_set_gemm_tn_config(3): 192 registers) leaves 128 registers per lane and SIMD: room for two waves of swiglu_bwd (52 VGPRs) or
one of ln_fwd_res (86) per SIMD.  This probe issues, from a common start event,
    stream A: one dW list (4M-B encoder or decoder layer)
    stream B: k launches of a streaming kernel (k chosen so that both sides take about as long alone)
and compares the wall time of the pair with the two alone - for the default GEMM form (no co-residency possible: the pair
serialises, up to tail filling) and for the 128-tile form.  Random bf16 data (DVFS: zeros clock higher).

    python tools/coresidency_probe.py [> profiles/r04_coresidency_probe.txt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-4m_amd"))
import torch  # noqa: E402
from fourm.hip import _lib as L  # noqa: E402
from fourm.hip import ops  # noqa: E402

dev = "cuda"
R, D, Hd = 256 * 128, 768, 2048
bf = torch.bfloat16


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(bf)


g_bf, act, h2, o, h1, hq, hc = rnd(R, D), rnd(R, Hd), rnd(R, D), rnd(R, D), rnd(R, D), rnd(R, D), rnd(R, D)
dgu, dqkv, dq, dkv = rnd(R, 2 * Hd), rnd(R, 3 * D), rnd(R, D), rnd(R, 2 * D)


def z(a, b):
    return torch.zeros(a, b, device=dev)


enc_jobs = [(g_bf, act, z(D, Hd), D, Hd, R), (dgu[:, :Hd], h2, z(Hd, D), Hd, D, R), (dgu[:, Hd:], h2, z(Hd, D), Hd, D, R),
            (g_bf, o, z(D, D), D, D, R), (dqkv, h1, z(3 * D, D), 3 * D, D, R)]
dec_jobs = enc_jobs + [(g_bf, o, z(D, D), D, D, R), (dq, hq, z(D, D), D, D, R), (dkv, hc, z(2 * D, D), 2 * D, D, R)]

# streaming partners
da, gu, dgu_out = rnd(R, Hd), rnd(R, 2 * Hd), torch.empty(R, 2 * Hd, device=dev, dtype=bf)
x32, delta, x_out, y_bf = torch.randn(R, D, device=dev), rnd(R, D), torch.empty(R, D, device=dev), torch.empty(R, D, device=dev, dtype=bf)
lnw = torch.ones(D, device=dev)
mu, rs = torch.zeros(R, device=dev), torch.ones(R, device=dev)
g32, dx_bf, dwln = torch.randn(R, D, device=dev), torch.empty(R, D, device=dev, dtype=bf), torch.zeros(D, device=dev)
big = torch.randn(360_000_000 // 4, device=dev)
ss = torch.zeros(1, device=dev)


def s_swiglu():
    ops.swiglu_bwd(da, gu, dgu_out, Hd, Hd, R=R)


def s_lnfwd():
    ops.layernorm_fwd(x32, lnw, None, y_bf, mu, rs, eps=1e-6, R=R, delta=delta, x_out=x_out)


def s_lnbwd():
    ops.layernorm_bwd(delta, x32, lnw, mu, rs, g32, dres=g32, dx_bf16=dx_bf, dw=dwln, R=R)


def s_sumsq():
    ops.sumsq(big, ss)


side = torch.cuda.Stream()


def run_pair(gemm, stream_fn, k, iters=10):
    """-> (gemm alone, k streaming launches alone, both from a common start) in microseconds."""
    main = torch.cuda.current_stream()

    def both():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for _ in range(k):
                stream_fn()
        gemm()
        main.wait_stream(side)

    def only_stream():
        for _ in range(k):
            stream_fn()

    out = []
    for fn in (gemm, only_stream, both):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) / iters * 1e3)
    return out


def trace_mode():
    """For `rocprofv3 --kernel-trace`: three 'pair' rounds per dW form with swiglu_bwd as the partner, nothing else, so that the
    kernel start / end timestamps show whether the two streams' kernels were resident at the same time (tools/coresidency_trace.py)."""
    main_s = torch.cuda.current_stream()
    for cfg in (1, 3):
        L.lib.fm_set_gemm_tn_config(cfg)
        for _ in range(4):
            side.wait_stream(main_s)
            with torch.cuda.stream(side):
                for _ in range(4):
                    s_swiglu()
            ops.gemm_tn_multi(enc_jobs)
            main_s.wait_stream(side)
            torch.cuda.synchronize()
    L.lib.fm_set_gemm_tn_config(1)


def main():
    if "--trace" in sys.argv:
        return trace_mode()
    print(f"# {torch.cuda.get_device_name(0)}; R={R}; times in us; 'pair' = both issued from a common start on two streams")
    print(f"{'dW form':28s} {'list':8s} {'partner':12s} {'k':>2s} {'gemm':>8s} {'stream':>8s} {'sum':>8s} {'pair':>8s} {'pair/sum':>8s} {'hidden':>7s}")
    partners = (("swiglu_bwd", s_swiglu, 4), ("ln_fwd_res", s_lnfwd, 8), ("ln_bwd", s_lnbwd, 6), ("sumsq", s_sumsq, 2))
    for cfg, cname in ((1, "256x256 K32 (256 VGPR)"), (3, "128x256 K64 (192 VGPR)")):
        L.lib.fm_set_gemm_tn_config(cfg)
        for lname, jobs in (("encoder", enc_jobs), ("decoder", dec_jobs)):
            for pname, fn, k in partners:
                kk = k if lname == "encoder" else k + k // 2
                g, s, p = run_pair(lambda: ops.gemm_tn_multi(jobs), fn, kk)
                print(f"{cname:28s} {lname:8s} {pname:12s} {kk:2d} {g:8.1f} {s:8.1f} {g + s:8.1f} {p:8.1f} {p / (g + s):8.3f} {(g + s - p) / min(g, s):7.2f}")
    L.lib.fm_set_gemm_tn_config(1)


if __name__ == "__main__":
    main()
