#!/bin/bash
# round 3, GPU call 3: sharding rewrite (overlap + owned heads) tests, per-rank critical path, more head-gating A/B
mkdir -p gpurun_out/r3c3
O=gpurun_out/r3c3
timeout 1500 python -m pytest tests/test_sharding_gpu.py tests/test_large_strips_gpu.py tests/test_sharding_multiproc_gpu.py \
   tests/test_stylize_sharded_gpu.py "tests/test_hot_path_gpu.py::test_closure_with_six_decades_of_channel_scales" -q -s > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -12 $O/pytest.log; grep "^\[strips\]" $O/pytest.log | grep -v "features\[" | head -30; grep "spread256\|channel-scale" $O/pytest.log | head
for v in "1 1" "0 0" "0 1" "2 1"; do set -- $v
  ST_STRIP_OVERLAP=$1 ST_STRIP_NS_OWNER=$2 timeout 300 python tools/strip_bench.py 2896x2172 8 2>&1 | grep strip_bench | sed "s/^/overlap=$1 owner=$2 /"
done
for v in "1 1" "0 0"; do set -- $v
  ST_STRIP_OVERLAP=$1 ST_STRIP_NS_OWNER=$2 timeout 300 python tools/strip_bench.py 2048 4 2>&1 | grep strip_bench | sed "s/^/overlap=$1 owner=$2 /"
done
ST_STRIP_OVERLAP=1 ST_STRIP_NS_OWNER=1 timeout 300 python tools/strip_bench.py 2048 8 2>&1 | grep strip_bench
tools/heads_ab.sh "512 256" 2 "base:" "g4:ST_HEAD_GATE=4" "g6:ST_HEAD_GATE=6" "g4h:ST_HEAD_GATE=4 ST_NS_F16_FWD_HEADS=8" "h8:ST_NS_F16_FWD_HEADS=8" > $O/heads_ab.log 2>&1
cat $O/heads_ab.log
for g in 4 6; do
  ST_HEAD_GATE=$g ST_AMD_TIMELINE=1 timeout 120 python bench.py --no-extra --no-cpu-baseline --size 512 --steps 40 --warmup 10 2> $O/timeline_g${g}_512.log > /dev/null
  echo "== gate $g size 512"; grep timeline $O/timeline_g${g}_512.log | tail -3
done
