cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof2048; mkdir -p $R/gpurun_out/prof2048
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof2048 -o b --output-format csv -- python $R/bench.py --size 2048 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof2048/log 2>&1
tail -1 $R/gpurun_out/prof2048/log | cut -c1-140
