# Round-end evidence: full GPU test suite, the default bench line, rocprofv3 kernel stats of the 512^2 and 2048^2 steps,
# PMC traffic of the conv launches (two separate --pmc passes, no trace domains besides --kernel-trace).
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/round
(timeout 1200 python -m pytest tests -q -m gpu -n 4 -rA --durations=10 2>&1) > gpurun_out/round/pytest.log 2>&1; tail -4 gpurun_out/round/pytest.log   # (pytest-xdist: 77 s instead of 282 s on one GPU)
(timeout 600 python bench.py) > gpurun_out/round/bench.log 2> gpurun_out/round/bench.err; tail -1 gpurun_out/round/bench.log | cut -c1-400
cd /tmp && export TMPDIR=/tmp
for sz in 512 2048; do
  rm -rf $R/gpurun_out/round/prof$sz; mkdir -p $R/gpurun_out/round/prof$sz
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/round/prof$sz -o b --output-format csv -- python $R/bench.py --size $sz --steps 18 --warmup 3 --no-extra --no-cpu-baseline > $R/gpurun_out/round/prof$sz/log.txt 2>&1
done
cd $R; ls gpurun_out/round/prof512 | head -3
bash tools/pmc_traffic.sh > gpurun_out/round/pmc.log 2>&1; python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic gpurun_out/round/pmc_traffic_conv.json 2>> gpurun_out/round/pmc.log | cut -c1-400
python tools/pmc_per_launch.py gpurun_out/pmc_traffic > gpurun_out/round/pmc_per_launch_512.txt 2>> gpurun_out/round/pmc.log
SIZE=2048 bash tools/pmc_traffic.sh >> gpurun_out/round/pmc.log 2>&1; SIZE=2048 python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic gpurun_out/round/pmc_traffic_conv_2048.json 2>> gpurun_out/round/pmc.log | cut -c1-400
python tools/pmc_per_launch.py gpurun_out/pmc_traffic > gpurun_out/round/pmc_per_launch_2048.txt 2>> gpurun_out/round/pmc.log
for cfg in "2048 2" "2048 4" "2048 8" "2896x2172 8"; do timeout 500 python tools/strip_bench.py $cfg 2>&1 | grep strip_bench; done > gpurun_out/round/strip_bench.txt
ST_AMD_TIMELINE=1 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>&1 | grep timeline | tail -3 > gpurun_out/round/timeline512.txt
(ST_FABRIC_SELF_HALO=1 ST_FABRIC_FORCE_COLLECTIVES=1 timeout 200 python tools/fabric_host_time.py 272 2896; ST_FABRIC_SELF_HALO=1 ST_FABRIC_FORCE_COLLECTIVES=1 timeout 200 python tools/fabric_host_time.py 256 2048) 2>&1 | grep "^\[fabric\]" > gpurun_out/round/fabric_host_time.txt
timeout 120 python tools/stylize_breakdown.py 2>&1 | grep breakdown > gpurun_out/round/stylize_breakdown.txt
