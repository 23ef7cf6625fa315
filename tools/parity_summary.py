"""Table of the parity numbers the GPU tests recorded (tests/parity_log.py -> gpurun_out/parity.jsonl):
    python tools/parity_summary.py [gpurun_out/parity.jsonl] > profiles/rNN_parity_summary.txt"""
import json
import sys


def flat(prefix, v, out):
    if isinstance(v, dict):
        for k, x in v.items():
            flat(f"{prefix}.{k}" if prefix else k, x, out)
    elif isinstance(v, float):
        out.append(f"{prefix}={v:.3g}")
    elif isinstance(v, list):
        if len(v) <= 6:
            out.append(f"{prefix}=[" + ", ".join(f"{x:.6g}" if isinstance(x, float) else str(x) for x in v) + "]")
    else:
        out.append(f"{prefix}={v}")


def main(path):
    last = {}
    for line in open(path):
        d = json.loads(line)
        last[(d["test"], json.dumps(d.get("case", d.get("name", ""))))] = d          # the newest record of a (test, case) wins
    for (test, _), d in sorted(last.items()):
        items = []
        body = {k: v for k, v in d.items() if k != "test"}
        pm = body.pop("per_modality", None)
        flat("", body, items)
        if pm:      # logits: worst modality per column
            cols = sorted({c for m in pm.values() for c in m})
            items += [f"worst_{c}={max(m[c] for m in pm.values()):.3g}" for c in cols]
        print(f"{test:34s} " + "  ".join(items))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity.jsonl")
