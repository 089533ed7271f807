mkdir -p gpurun_out/r5m
for l2 in 0 1; do echo "== ST_NS_CHAIN_L2=$l2"; ST_NS_CHAIN_L2=$l2 timeout 200 python tools/ns_chain_bench.py 2>&1 | grep -v "^/opt" | grep -E "^\| (512|256|64) \|" ; done > gpurun_out/r5m/chain_l2.txt 2>&1
ST_NS_CHAIN_L2=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -k "persistent_chain" -q -s 2>&1 | tail -12 >> gpurun_out/r5m/chain_l2.txt
for rep in 1 2; do for cfg in "0 0" "4 0" "4 1"; do set -- $cfg; for sz in 512 128; do echo "mask=$1 l2=$2 size=$sz"; ST_NS_CHAIN=$1 ST_NS_CHAIN_SYM=0 ST_NS_CHAIN_L2=$2 timeout 120 python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 --size $sz 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  it/s', round(d['value'],1), [round(v,1) for v in d['value_regions']], 'loss', d['final_loss'])
"; done; done; done > gpurun_out/r5m/bench_ab.txt 2>&1
