# round-2 GPU call C: NS chains after the scaled-residual fix (timing + parity), launch modes, kernel trace of the chains
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 60 python tools/ns_bench.py > gpurun_out/r2c_ns.log 2>&1; cat gpurun_out/r2c_ns.log
timeout 200 python tools/launch_modes.py 128 256 512 > gpurun_out/r2c_modes.log 2>&1; cat gpurun_out/r2c_modes.log
(timeout 200 python -m pytest tests/test_kernels_gpu.py tests/test_hot_path_gpu.py -q -m gpu -rA -k "sqrtm or live_oracle or goldens or lyapunov or lbfgs" --timeout 200 2>&1) > gpurun_out/r2c_pytest.log 2>&1; tail -5 gpurun_out/r2c_pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2c_prof; mkdir -p $R/gpurun_out/r2c_prof
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2c_prof -o ns --output-format csv -- python $R/tools/ns_bench.py > $R/gpurun_out/r2c_prof/log.txt 2>&1
cd $R; ls gpurun_out/r2c_prof | head; python tools/prof_summary.py gpurun_out/r2c_prof/ns_kernel_stats.csv | head -30
