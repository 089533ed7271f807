R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
(timeout 500 python -m pytest tests -q -m gpu -rA --timeout 300 2>&1) > gpurun_out/r2i_pytest.log 2>&1; tail -6 gpurun_out/r2i_pytest.log
(timeout 170 python bench.py) > gpurun_out/r2i_bench.log 2>&1; tail -c 1500 gpurun_out/r2i_bench.log | cut -c1-1500
