# round-2 GPU call D: per-head graphs (launch modes), parity subset, bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 300 python tools/launch_modes.py 128 256 512 > gpurun_out/r2d_modes.log 2>&1; cat gpurun_out/r2d_modes.log
(timeout 250 python -m pytest tests/test_hot_path_gpu.py -q -m gpu -rA --timeout 200 2>&1) > gpurun_out/r2d_pytest.log 2>&1; tail -8 gpurun_out/r2d_pytest.log
(timeout 170 python bench.py --no-cpu-baseline) > gpurun_out/r2d_bench.log 2>&1; tail -c 2500 gpurun_out/r2d_bench.log
