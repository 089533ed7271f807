#!/bin/bash
# avg_launch_us of the HBM-bound side kernels (bench.py's roofline_hbm, HIP events around the launches) per size and variant:
#   tools/hbm_kernels_ab.sh "128 512" "new:" "old:ST_CONV1_CO_SPLIT=0 ST_CONV1_DGRAD_SPLIT=0"
SIZES="$1"; shift
cd "$(dirname "$0")/.."
for size in $SIZES; do
  for spec in "$@"; do
    name="${spec%%:*}"; envs="${spec#*:}"
    env $envs timeout 120 python bench.py --no-extra --no-cpu-baseline --size "$size" --steps 40 --warmup 10 2>/dev/null |
      python -c "
import sys, json
d = json.loads(sys.stdin.readline())
h = d.get('roofline_hbm', {})
print('[hbm_ab] size $size $name: %.1f it/s | ' % d['value'] + ' | '.join('%s %.1f us' % (k.split(' (')[0], v['avg_launch_us']) for k, v in h.items()))"
  done
done
