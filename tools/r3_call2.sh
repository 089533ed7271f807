#!/bin/bash
# round 3, GPU call 2: large-size parity tests after the XL halo-flag fix + head gating A/B + timelines
mkdir -p gpurun_out/r3c2
O=gpurun_out/r3c2
timeout 1200 python -m pytest tests/test_large_strips_gpu.py "tests/test_hot_path_gpu.py::test_closure_against_reference_goldens_at_baseline_sizes" -q -s > $O/pytest_new.log 2>&1
echo "pytest exit $?"; tail -15 $O/pytest_new.log; grep "^\[strips\]" $O/pytest_new.log | head -40
for g in 0 1 3 9; do
  for size in 512 256; do
    ST_HEAD_GATE=$g ST_AMD_TIMELINE=1 timeout 120 python bench.py --no-extra --no-cpu-baseline --size $size --steps 40 --warmup 10 2> $O/timeline_g${g}_$size.log > /dev/null
    echo "== gate $g size $size"; grep timeline $O/timeline_g${g}_$size.log | tail -3
  done
done
tools/heads_ab.sh "512 256 128" 2 "base:" "g1:ST_HEAD_GATE=1" "g2:ST_HEAD_GATE=2" "g3:ST_HEAD_GATE=3" "g8:ST_HEAD_GATE=8" "g10:ST_HEAD_GATE=10" \
   "g1f:ST_HEAD_GATE=1 ST_NS_F16_FWD=1" > $O/heads_ab.log 2>&1
cat $O/heads_ab.log
