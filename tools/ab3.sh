#!/bin/bash
# same-box A/B/C... of the whole train step between several environment settings:  tools/ab3.sh "A=1" "A=2" "A=3"
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for setting in "$@"; do
    eval "env $setting python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); kb=d.get('kernel_breakdown_ms_per_step',{})
print('rep $rep  [$setting]  %.2f ms  ' % d['ms_per_step'] + '  '.join('%s %.2f' % (k, v) for k, v in list(kb.items())[:4]))"
  done
done
