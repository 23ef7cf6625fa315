#!/bin/bash
# same-box A/B of the 4M-L mod21 train step (BASELINE configs[3]) between two environment settings:  tools/ab_env21.sh "FOURM_NT4=1" "FOURM_NT4=3"
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
  for i in 1 2; do
    eval "env \${$i} python bench.py --mods mod21 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); kb=d.get('kernel_breakdown_ms_per_step',{})
print('rep $rep  [${!i}]  %.2f ms  ' % d['ms_per_step'] + '  '.join('%s %.2f' % (k, v) for k, v in list(kb.items())[:9]))"
  done
done
