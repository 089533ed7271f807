OUT=gpurun_out/r5p; mkdir -p $OUT
{ for cfg in "0 0" "1 2" "5 2" "1 0" "0 0"; do set -- $cfg; echo "mask=$1 l2=$2"; for sz in 512 128; do ST_AMD_TIMELINE=1 ST_NS_CHAIN=$1 ST_NS_CHAIN_SYM=0 ST_NS_CHAIN_L2=$2 timeout 120 python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline --no-pmc --size $sz 2>&1 | grep -E "timeline|^\{" | python -c "
import sys,json
last=None
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  $sz it/s', round(d['value'],1), [round(v,1) for v in d['value_regions']], 'loss', d['final_loss'])
    elif 'forward end' in l: last=l.strip()
print('  ', last)
"; done; done; } > $OUT/shallow.txt 2>&1
