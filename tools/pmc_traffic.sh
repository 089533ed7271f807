# HBM-side traffic of the conv launches of the default bench (two separate PMC passes, MI355X_MICROARCH.md "HBM").
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_traffic; mkdir -p $R/gpurun_out/pmc_traffic
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_traffic/f -o f --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_traffic/w -o w --output-format csv -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extra > /dev/null 2>&1
ls $R/gpurun_out/pmc_traffic/f $R/gpurun_out/pmc_traffic/w
