R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python tools/launch_modes.py 128 256 512 > gpurun_out/r2g_modes.log 2>&1; cat gpurun_out/r2g_modes.log
(timeout 400 python -m pytest tests -q -m gpu -rA --timeout 300 -x 2>&1) > gpurun_out/r2g_pytest.log 2>&1; tail -6 gpurun_out/r2g_pytest.log
(timeout 170 python bench.py) > gpurun_out/r2g_bench.log 2>&1; tail -c 2800 gpurun_out/r2g_bench.log
