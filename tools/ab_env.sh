#!/bin/bash
# same-box A/B of the whole train step between two environment settings:  tools/ab_env.sh "FOURM_TN_CONFIG=1" "FOURM_TN_CONFIG=4"
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do
  for i in 1 2; do
    eval "env \${$i} python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null" | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); kb=d.get('kernel_breakdown_ms_per_step',{})
print('rep $rep  [${!i}]  %.2f ms  ' % d['ms_per_step'] + '  '.join('%s %.2f' % (k, v) for k, v in kb.items()))"
  done
done
