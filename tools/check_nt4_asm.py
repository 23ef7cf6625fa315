#!/usr/bin/env python3
"""gemm_nt4.hip and gemm_tn4.hip keep 16 accumulator fragments in hand-named AGPRs the compiler does not know about.  That is only sound while
the compiler itself never touches the AGPR file in those kernels: compile the files to assembly and check, per kernel, that no
instruction outside the ASMSTART / ASMEND blocks references an AGPR and that nothing goes to scratch.  Run by tests/test_model_cpu.py
(hipcc cross-compiles without a GPU)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


FILES = (("gemm_nt4.hip", "gemm_nt4_kernel"), ("gemm_tn4.hip", "gemm_tn4_multi_kernel"))


def compile_to_asm(tmp, fname="gemm_nt4.hip"):
    src = os.path.join(ROOT, "ml-4m_amd", "csrc", fname)
    out = os.path.join(tmp, fname.replace(".hip", ".s"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-value", "-x", "hip", "--cuda-device-only", "-S",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ml-4m_amd", "csrc"), src, "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    return out


def check(path, kernel="gemm_nt4_kernel"):
    agpr = re.compile(r"(?<![A-Za-z0-9_.])a(\d+|\[)")
    problems, kernels = [], {}
    name, in_asm = None, False
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_ZN\S*" + kernel + r"\S*):", line)
        if m:
            name = m.group(1); kernels[name] = dict(mfma=0, asm_agpr=0); in_asm = False
            continue
        if name is None:
            continue
        if ".end_amdhsa_kernel" in line or line.startswith(".Lfunc_end"):
            name = None
            continue
        code = line.split(";")[0] if not line.lstrip().startswith(";;#") else line
        if ";;#ASMSTART" in line:
            in_asm = True; continue
        if ";;#ASMEND" in line:
            in_asm = False; continue
        if in_asm:
            kernels[name]["mfma"] += "v_mfma" in code
            kernels[name]["asm_agpr"] += bool(agpr.search(code))
            continue
        if agpr.search(code) or "v_accvgpr" in code:
            problems.append(f"{name}: line {ln}: AGPR outside an asm block: {line.strip()}")
        if "scratch_" in code:
            problems.append(f"{name}: line {ln}: scratch access: {line.strip()}")
    return kernels, problems


if __name__ == "__main__":
    bad = False
    with tempfile.TemporaryDirectory() as tmp:
        for fname, kernel in FILES:
            kernels, problems = check(compile_to_asm(tmp, fname), kernel)
            for k, v in kernels.items():
                print(k, v)
            for p in problems[:40]:
                print("PROBLEM", p)
            bad = bad or bool(problems) or not kernels
    sys.exit(1 if bad else 0)
