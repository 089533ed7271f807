#!/bin/bash
mkdir -p gpurun_out/r3c7
O=gpurun_out/r3c7
timeout 1200 python -m pytest tests/test_sharding_gpu.py tests/test_large_strips_gpu.py tests/test_sharding_multiproc_gpu.py -q > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -6 $O/pytest.log
timeout 300 python tools/strip_bench.py 2896x2172 8 2>&1 | grep strip_bench | tee $O/strip_2896x8.log
timeout 300 python tools/strip_bench.py 2048 4 2>&1 | grep strip_bench | tee $O/strip_2048x4.log
timeout 300 python tools/strip_bench.py 2048 8 2>&1 | grep strip_bench | tee $O/strip_2048x8.log
ST_STRIP_OVERLAP=0 timeout 300 python tools/strip_bench.py 2896x2172 8 2>&1 | grep strip_bench | sed 's/^/overlap=0 /' | tee $O/strip_2896x8_nooverlap.log
