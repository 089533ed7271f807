# rocprofv3 kernel trace of a few bench.py steps -> gpurun_out/trace/t_kernel_trace.csv (analyse with tools/trace_iter.py)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/trace; mkdir -p $R/gpurun_out/trace
rocprofv3 --kernel-trace -d $R/gpurun_out/trace -o t --output-format csv -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/trace/bench.log 2>&1
tail -1 $R/gpurun_out/trace/bench.log | cut -c1-160
ls -la $R/gpurun_out/trace
