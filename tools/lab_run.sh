#!/bin/bash
cd "$(dirname "$0")/.."
python tools/ar_bench.py 16 64 2>&1 | grep -v Warn | tail -4
python tools/ar_bench.py 1 64 2>&1 | grep -v Warn | tail -1
