#!/bin/bash
# round 3, GPU call 1: new parity tests + XCD probe + head-stream placement A/B + timelines
mkdir -p gpurun_out/r3c1
O=gpurun_out/r3c1
python - > $O/probe.log 2>&1 <<'PY'
import sys; sys.path.insert(0, 'style-transfer-pytorch_amd')
from style_transfer import _hip
for s in (15, 240, 1, 2, 3, 192, 63):
    print('xcc set 0x%02x ->' % s, _hip.xcc_stream_probe(s))
PY
cat $O/probe.log
timeout 900 python -m pytest tests/test_large_strips_gpu.py "tests/test_hot_path_gpu.py::test_closure_against_reference_goldens_at_baseline_sizes" -x -q -s > $O/pytest_new.log 2>&1
echo "pytest exit $?"; tail -5 $O/pytest_new.log
for size in 512 256; do
  ST_AMD_TIMELINE=1 timeout 120 python bench.py --no-extra --no-cpu-baseline --size $size --steps 40 --warmup 10 2> $O/timeline_base_$size.log | cut -c1-200
  ST_HEAD_XCC4=15 ST_HEAD_XCC3=240 ST_AMD_TIMELINE=1 timeout 120 python bench.py --no-extra --no-cpu-baseline --size $size --steps 40 --warmup 10 2> $O/timeline_half_$size.log | cut -c1-200
done
tools/heads_ab.sh "512 256 128" 2 "base:" "half:ST_HEAD_XCC4=15 ST_HEAD_XCC3=240" "h6_2:ST_HEAD_XCC4=63 ST_HEAD_XCC3=192" \
   "half_all:ST_HEAD_XCC4=15 ST_HEAD_XCC3=240 ST_HEAD_XCC012=240" "only4:ST_HEAD_XCC4=15" "only3:ST_HEAD_XCC3=240" \
   "two:ST_HEAD_XCC4=3 ST_HEAD_XCC3=12 ST_HEAD_XCC012=240" "one:ST_HEAD_XCC4=1 ST_HEAD_XCC3=2 ST_HEAD_XCC012=252" \
   "f16fwd:ST_NS_F16_FWD=1" > $O/heads_ab.log 2>&1
cat $O/heads_ab.log
grep timeline $O/timeline_*_512.log | tail -8
