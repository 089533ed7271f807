#!/bin/bash
mkdir -p gpurun_out/r3c11
O=gpurun_out/r3c11
for mode in 0 2; do
 for padv in 0 10 20 30 1 11 21 2 12; do
  ST_HEAD_STREAMS=$mode ST_HEAD_STREAM_PAD=$padv ST_AMD_TIMELINE=1 timeout 120 python bench.py --no-extra --no-cpu-baseline --size 512 --steps 50 --warmup 10 2> $O/tl_${mode}_${padv}.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('mode $mode pad $padv: %.1f it/s' % d['value'], end='  ')"
  grep "relu4_1 head" $O/tl_${mode}_${padv}.log | tail -1 | cut -c12-140
 done
done
