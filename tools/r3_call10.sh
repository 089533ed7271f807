#!/bin/bash
mkdir -p gpurun_out/r3c10
O=gpurun_out/r3c10
tools/heads_ab.sh "512 256 128" 3 "base:" "s1:ST_HEAD_STREAMS=1" "s2:ST_HEAD_STREAMS=2" > $O/heads_ab.log 2>&1
cat $O/heads_ab.log
for m in 1 2; do
  ST_HEAD_STREAMS=$m ST_AMD_TIMELINE=1 timeout 120 python bench.py --no-extra --no-cpu-baseline --size 512 --steps 40 --warmup 10 2> $O/tl_s${m}_512.log > /dev/null
  echo "== streams mode $m"; grep timeline $O/tl_s${m}_512.log | tail -3
done
