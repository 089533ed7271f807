# A/B of the XCD-aware workgroup order of the heads' Gram and 1 x 1 kernels (ST_XCD_REMAP, round 6): kernel durations from a
# rocprofv3 kernel trace of the isolated launches, then the whole step.   gpurun -- bash tools/xcd_remap_ab.sh [size]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SZ=${1:-2048}
mkdir -p $R/gpurun_out/xcd
for V in 0 1; do
  for T in gram_bench conv1x1_bench; do
    rm -rf /tmp/xcdp; ST_XCD_REMAP=$V rocprofv3 --kernel-trace --stats -d /tmp/xcdp -o x --output-format csv -- python $R/tools/$T.py $SZ > /tmp/xcdp.log 2>&1
    F=$(find /tmp/xcdp -name "*kernel_stats.csv" | head -1)
    echo "== ST_XCD_REMAP=$V $T $SZ"
    python - $F <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'gram_partial' in n or 'conv1x1_f16' in n or 'gram_finalize' in n:
        print(f"  {n.split('(')[0][:60]:62s} calls {r['Calls']:>4s} avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}  max {float(r['MaxNs']) / 1e3:8.1f}")
P
  done
done
for V in 0 1 0 1; do
  echo "== ST_XCD_REMAP=$V bench $SZ"; ST_XCD_REMAP=$V python $R/bench.py --no-extra --no-cpu-baseline --no-pmc --size $SZ --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  it/s', round(d['value'],2), 'regions', [round(x,2) for x in d['value_regions']])"
done
