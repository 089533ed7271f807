mkdir -p gpurun_out/r5n
for rep in 1 2; do for cfg in "0 0" "1 1" "3 1" "2 1" "1 0"; do set -- $cfg; for sz in 512 256 128; do echo "mask=$1 delay=$2 size=$sz"; ST_NS_CHAIN=$1 ST_NS_CHAIN_SYM=0 ST_NS_CHAIN_DELAY=$2 ST_AMD_TIMELINE=$((sz==512)) timeout 120 python bench.py --no-extra --no-cpu-baseline --steps 40 --warmup 10 --size $sz 2>&1 | grep -E "^\{|forward end" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  it/s', round(d['value'],1), [round(v,1) for v in d['value_regions']], 'loss', d['final_loss'])
    else: last=l.strip()
try: print('  ', last)
except NameError: pass
"; done; done; done > gpurun_out/r5n/bench_ab.txt 2>&1
