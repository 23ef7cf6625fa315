#!/usr/bin/env python3
"""The CPU leg of bench.py, both ways on the same cores (build container only: needs /root/reference): the UNMODIFIED upstream
fourm.models.fm.FourM and the oracle port (oracle/fourm_oracle.py), 4M-B mod7, batch 8, fp32, forward + backward + AdamW, in child
processes, interleaved.  Writes profiles/r05_cpu_port_vs_reference.{txt,json}: the conversion bench.py attaches to a `kind: "port"`
record on machines where the reference tree does not exist (the GPU boxes).
    python tools/cpu_port_vs_reference.py [rounds]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rows = []
for r in range(rounds):
    for kind in ("reference", "port"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-worker", "--cpu-baseline-kind", kind], capture_output=True, text=True,
                             env={**os.environ, "HIP_VISIBLE_DEVICES": "", "CUDA_VISIBLE_DEVICES": ""})
        rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert rec["kind"] == kind, rec
        rows.append((r, kind, rec))
        print(f"round {r}  {kind:9s}  {rec['value']:8.1f} tokens/s  ({rec['sample'].rsplit(',', 1)[-1].strip()})", flush=True)
med = lambda xs: sorted(xs)[len(xs) // 2]
ref = med([x[2]["value"] for x in rows if x[1] == "reference"])
port = med([x[2]["value"] for x in rows if x[1] == "port"])
rec0 = rows[0][2]
summary = {"reference_tokens_per_s": ref, "port_tokens_per_s": port, "port_over_reference": port / ref, "cores": rec0["cores"], "host_cores": rec0["host_cores"],
           "cpu": rec0.get("cpu", ""), "rounds": rounds, "workload": "4M-B mod7, batch 8, 128+128 tokens, fp32, fwd+bwd+AdamW, median of 3 steps per run"}
with open(os.path.join(ROOT, "profiles", "r05_cpu_port_vs_reference.txt"), "w") as f:
    f.write("# tools/cpu_port_vs_reference.py: unmodified upstream FourM vs the oracle port, same cores, child processes, interleaved\n")
    for r, kind, rec in rows:
        f.write(f"round {r}  {kind:9s}  {rec['value']:8.1f} tokens/s  {rec['sample']}\n")
    f.write(json.dumps(summary) + "\n")
json.dump(summary, open(os.path.join(ROOT, "profiles", "r05_cpu_port_vs_reference.json"), "w"))
print(json.dumps(summary))
