R=$GRAFT_REPO_ROOT; cd $R
bash tools/pmc_traffic.sh > /dev/null 2>&1
python tools/pmc_traffic_summary.py gpurun_out/pmc_traffic gpurun_out/r02_pmc_traffic_conv.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2q_tv; mkdir -p $R/gpurun_out/r2q_tv
timeout 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r2q_tv -o tv --output-format csv -- python $R/tools/side_bench.py > $R/gpurun_out/r2q_tv/log.txt 2>&1
cd $R; python tools/prof_summary.py gpurun_out/r2q_tv/tv_kernel_stats.csv | head -8
grep "tv_kernel" gpurun_out/r2q_tv/tv_kernel_trace.csv | awk -F, '{gsub(/"/,""); print $(NF-2), $(NF-12)-$(NF-13)}' | sort | uniq -c | sort -k2 -n | head -20
