#!/bin/bash
mkdir -p gpurun_out/r3c9
O=gpurun_out/r3c9
for m in 0 15 8 7; do
  for size in 512 256; do
    ST_ABLATE_HEADS=$m ST_AMD_TIMELINE=1 timeout 120 python bench.py --no-extra --no-cpu-baseline --size $size --steps 40 --warmup 10 2> $O/tl_m${m}_$size.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ablate heads mask $m size $size: %.1f it/s' % d['value'])"
    grep timeline $O/tl_m${m}_$size.log | tail -3
  done
done
python tools/ns_bench.py 2>&1 | tail -12
