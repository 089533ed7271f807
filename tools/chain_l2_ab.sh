# A/B of the persistent chain kernel's operand path (ST_NS_CHAIN_L2: 0 memory side, 1 L2 + acquire per barrier, 2 L2 + a matrix
# of its own for every iterate): isolated chain times, the operator test, then bench.py at 512^2 / 128^2 with relu5_1's head on
# the kernel (every tile).      gpurun -- bash tools/chain_l2_ab.sh [modes, default "0 1 2"]
MODES=${1:-"0 1 2"}
OUT=gpurun_out/r5p
mkdir -p $OUT
for l2 in $MODES; do echo "== ST_NS_CHAIN_L2=$l2"; ST_NS_CHAIN_L2=$l2 timeout 200 python tools/ns_chain_bench.py 2>&1 | grep -v "^/opt" | grep -E "^\| (512|256|64) \|" ; done > $OUT/chain_l2.txt 2>&1
for l2 in $MODES; do echo "== ST_NS_CHAIN_L2=$l2"; ST_NS_CHAIN_L2=$l2 timeout 300 python -m pytest tests/test_kernels_gpu.py -k "persistent_chain" -q -s 2>&1 | tail -12; done >> $OUT/chain_l2.txt
run() { echo "mask=$1 l2=$2 size=$3"; ST_NS_CHAIN=$1 ST_NS_CHAIN_SYM=0 ST_NS_CHAIN_L2=$2 timeout 120 python bench.py --no-extra --no-cpu-baseline --no-pmc --steps 40 --warmup 10 --size $3 2>&1 | grep -E "^\{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  it/s', round(d['value'],1), [round(v,1) for v in d['value_regions']], 'loss', d['final_loss'])
"; }
for rep in 1 2; do for sz in 512 128; do run 0 0 $sz; for m in $MODES; do run 4 $m $sz; done; done; done > $OUT/bench_ab.txt 2>&1
