# Round-end evidence: rocprofv3 kernel stats of the default bench + it/s at the BASELINE scale list.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_final; mkdir -p $R/gpurun_out/prof_final
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_final -o bench --output-format csv -- python $R/bench.py --steps 15 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_final/bench_profiled.log 2>&1
cd $R
python bench.py --steps 100 --warmup 20 > gpurun_out/prof_final/bench512.json 2> gpurun_out/prof_final/bench512.err
for sz in 128 256 1024 2048; do
  st=60; [ $sz -ge 1024 ] && st=20
  python bench.py --size $sz --steps $st --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/prof_final/bench$sz.json
done
ls -la gpurun_out/prof_final; for f in gpurun_out/prof_final/bench*.json; do echo $f; cut -c1-150 $f; done
