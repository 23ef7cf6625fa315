"""Measured parity numbers of a GPU test run, appended as JSON lines to ``gpurun_out/parity.jsonl`` (or
``$FOURM_PARITY_LOG``) so that the figures DESIGN.md quotes are the ones the asserts saw (pytest -q swallows prints)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(test: str, **values):
    path = os.environ.get("FOURM_PARITY_LOG", os.path.join(ROOT, "gpurun_out", "parity.jsonl"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(json.dumps({"test": test, **values}) + "\n")
    except OSError:
        pass
