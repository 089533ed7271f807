# round-2 GPU call E: per-stream kernel timeline of one iteration at 128^2 and 512^2 (eager heads, so that streams stay visible)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for sz in 128 512; do
  rm -rf $R/gpurun_out/r2e_trace$sz; mkdir -p $R/gpurun_out/r2e_trace$sz
  ST_HEAD_GRAPH=0 timeout 150 rocprofv3 --kernel-trace -d $R/gpurun_out/r2e_trace$sz -o t --output-format csv -- python $R/bench.py --size $sz --steps 12 --warmup 4 --no-extra --no-cpu-baseline > $R/gpurun_out/r2e_trace$sz/log.txt 2>&1
  tail -c 300 $R/gpurun_out/r2e_trace$sz/log.txt
done
cd $R
for sz in 128 512; do python tools/trace_iter.py gpurun_out/r2e_trace$sz/t_kernel_trace.csv | head -12; done
