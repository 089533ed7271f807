#!/bin/bash
# round 3, GPU call 4: overlap-part op test (localise the 96x80 gradient error), range guard, fp16x6 chain, conv1_1 wide kernel
mkdir -p gpurun_out/r3c4
O=gpurun_out/r3c4
timeout 900 python -m pytest "tests/test_large_strips_gpu.py::test_conv_interior_plus_boundary_equals_one_launch" \
   "tests/test_hot_path_gpu.py::test_closure_with_six_decades_of_channel_scales" \
   "tests/test_kernels_gpu.py::test_sqrtm_diag_backward_and_fp16x3_chains" "tests/test_kernels_gpu.py::test_conv1_1_four_pixel_kernel_is_bit_identical" \
   "tests/test_kernels_gpu.py::test_trunk_forward_taps" -q -s > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -15 $O/pytest.log; grep "conv overlap\|NS chains n=512\|spread256\|channel-scale" $O/pytest.log | cut -c1-330
tools/heads_ab.sh "512 256 128" 2 "base:" "x6:ST_NS_F16X6_FWD=1" "f16:ST_NS_F16_FWD=1" "c1n:ST_CONV1_WIDE=0" > $O/heads_ab.log 2>&1
cat $O/heads_ab.log
for v in 1 0; do
ST_CONV1_WIDE=$v timeout 200 python bench.py --no-extra --no-cpu-baseline --size 2048 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('conv1_wide=$v 2048: %.2f it/s' % d['value'], {k:(round(v['achieved']),round(v['avg_launch_us'])) for k,v in d.get('roofline_hbm',{}).items()})"
done
