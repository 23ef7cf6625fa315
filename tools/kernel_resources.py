#!/usr/bin/env python3
"""Registers / LDS / occupancy of every gfx950 kernel, from hipcc's own remarks (no GPU needed).

    python tools/kernel_resources.py [file.hip ...] > profiles/rNN_kernel_resources.txt

Compiles the device side of each csrc/*.hip with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
VGPRs, AGPRs, SGPRs, waves per SIMD the allocation allows, STATIC LDS bytes per workgroup (the GEMM / attention kernels size their
LDS dynamically at launch: 160 KB for gemm_nt3 with its staged epilogue, 96 KB for gemm_tn_multi, 36 - 68 KB for attention), scratch.  The allocation granule is 8
registers per lane and waves/SIMD = min(8, 512 // alloc) (MI355X_MICROARCH.md, register files): a second kernel can share a
CU with a resident workgroup only inside what these numbers leave (512 registers per lane and SIMD, 160 KB of LDS, 32 waves)."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ml-4m_amd", "csrc")
KEYS = (("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"SGPRs"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("lds", r"LDS Size \[bytes/block\]"), ("scratch", r"ScratchSize \[bytes/lane\]"))


def remarks(src):
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value", "-x", "hip",
               "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src,
               "-o", os.path.join(tmp, "x.o")]
        if os.path.basename(src) == "sample.hip":
            cmd.append("-ffp-contract=off")
        return subprocess.run(cmd, capture_output=True, text=True).stderr


def main():
    files = sys.argv[1:] or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    files = [f if os.path.isabs(f) else os.path.join(CSRC, os.path.basename(f)) for f in files]
    with ThreadPoolExecutor(8) as ex:
        outs = list(ex.map(remarks, files))
    for f, txt in zip(files, outs):
        print(f"== {os.path.basename(f)}")
        rows = []
        for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
            name = b.split("\n")[0].strip().split(" ")[0]
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dn = re.sub(r"\(anonymous namespace\)::", "", dn)
            dn = re.sub(r"\((fmk::|unsigned|float|int|void|fm_|NTArgs|TNArgs|AttnArgs|TNMultiArgs|SelArgs|FoldArgs|fm_).*$", "", dn)
            dn = dn.replace("void ", "")
            vals = {}
            for k, pat in KEYS:
                m = re.search(pat + r": (\S+)", b)
                vals[k] = m.group(1) if m else "?"
            rows.append((dn, vals))
        for dn, v in rows:
            print(f"{dn[:96]:96s} vgpr {v['vgpr']:>3} agpr {v['agpr']:>3} sgpr {v['sgpr']:>3} waves/SIMD {v['occ']:>1} lds {v['lds']:>6} scratch {v['scratch']}")


if __name__ == "__main__":
    main()
