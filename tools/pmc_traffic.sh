# HBM-side traffic of the conv launches of the bench step (two separate PMC passes, MI355X_MICROARCH.md "HBM").
# SIZE=2048 tools/pmc_traffic.sh : another image size (default 512, the bench workload)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_traffic; mkdir -p $R/gpurun_out/pmc_traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_traffic/f -o f --output-format csv -- python $R/bench.py --size ${SIZE:-512} --steps 3 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_traffic/w -o w --output-format csv -- python $R/bench.py --size ${SIZE:-512} --steps 3 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
ls $R/gpurun_out/pmc_traffic/f $R/gpurun_out/pmc_traffic/w
