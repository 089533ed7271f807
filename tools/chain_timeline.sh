mkdir -p gpurun_out/r5p
for cfg in "0 0" "4 2" "4 0" "6 2"; do set -- $cfg; echo "mask=$1 l2=$2"; ST_AMD_TIMELINE=1 ST_NS_CHAIN=$1 ST_NS_CHAIN_SYM=0 ST_NS_CHAIN_L2=$2 timeout 120 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline --no-pmc 2>&1 | grep -E "timeline|^\{" | sed -E 's/^\{.*"value": ([0-9.]+).*/  it\/s \1/' | tail -4; done > gpurun_out/r5p/timeline.txt 2>&1
