# quick look at the persistent chain kernel's L2 / LDS-DMA form: operator test, isolated times, in-situ timeline + it/s
OUT=gpurun_out/r5p; mkdir -p $OUT
{ ST_NS_CHAIN_L2=2 timeout 300 python -m pytest tests/test_kernels_gpu.py -k "persistent_chain" -q -s 2>&1 | grep -E "parity|passed|failed|Error" | cut -c1-330
  ST_NS_CHAIN_L2=2 timeout 200 python tools/ns_chain_bench.py 2>&1 | grep -E "^\| (512|256|128|64) \| (one|persistent kernel)"
  for cfg in "0 0" "4 2" "6 2" "7 2"; do set -- $cfg; echo "mask=$1 l2=$2"; for sz in 512 128; do ST_AMD_TIMELINE=1 ST_NS_CHAIN=$1 ST_NS_CHAIN_SYM=0 ST_NS_CHAIN_L2=$2 timeout 120 python bench.py --steps 40 --warmup 10 --no-extra --no-cpu-baseline --no-pmc --size $sz 2>&1 | grep -E "timeline|^\{" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  $sz it/s', round(d['value'],1), [round(v,1) for v in d['value_regions']], 'loss', d['final_loss'])
    elif 'forward end' in l: print('  '+l.strip())
"; done; done; } > $OUT/quick.txt 2>&1
