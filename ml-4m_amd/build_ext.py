#!/usr/bin/env python3
"""Compile the gfx950 kernels into ml-4m_amd/fourm/_lib/libfourm_hip.so (in-tree, so the object travels
with the snapshot to the GPU box).  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "fourm", "_lib", "libfourm_hip.so")
SOURCES = ["api.cpp", "gemm.hip", "gemm_nt_flat.hip", "gemm_nt3.hip", "gemm_nt4.hip", "gemm_tn4.hip", "gemm_skinny.hip", "layernorm.hip", "attention.hip", "select_embed.hip", "loss.hip", "elementwise.hip", "vq.hip", "fp32_verify.hip", "sample.hip", "masking.hip", "unet.hip"]


# sample.hip: the sampler's determinism contract needs separately rounded fp32 multiplies and adds (no fused multiply-add)
EXTRA_FLAGS = {"sample.hip": ["-ffp-contract=off"]}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def up_to_date(srcs) -> bool:
    if not os.path.exists(OUT):
        return False
    t = os.path.getmtime(OUT)
    deps = srcs + [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_args.h"), os.path.join(CSRC, "agpr_mfma.h"), os.path.join(ROOT, "include", "fourm_hip.h"), __file__]
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and up_to_date(srcs):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objs = []
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= max(
                os.path.getmtime(s), os.path.getmtime(os.path.join(CSRC, "common.h")), os.path.getmtime(os.path.join(CSRC, "gemm_args.h")), os.path.getmtime(os.path.join(CSRC, "agpr_mfma.h")),
                os.path.getmtime(os.path.join(ROOT, "include", "fourm_hip.h"))):
            continue
        cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value",
               "-x", "hip", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-c", s, "-o", o] + EXTRA_FLAGS.get(os.path.basename(s), [])
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {s}")
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
