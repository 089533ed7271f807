cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/stripprof; mkdir -p $R/gpurun_out/stripprof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stripprof -o s --output-format csv -- python $R/tools/strip_bench.py 512 1 > $R/gpurun_out/stripprof/log 2>&1
tail -1 $R/gpurun_out/stripprof/log
