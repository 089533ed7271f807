# A/B of the fat convolution kernel in the iteration (ST_CONV_FAT=0 / 1), separate processes, interleaved.   gpurun -- bash tools/fat_ab.sh "1024 2048"
R=$GRAFT_REPO_ROOT
for SZ in ${1:-1024 2048}; do
  for V in 0 1 0 1; do
    echo "== ST_CONV_FAT=$V bench $SZ"; ST_CONV_FAT=$V python $R/bench.py --no-extra --no-cpu-baseline --no-pmc --size $SZ --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  it/s', round(d['value'],2), 'regions', [round(x,2) for x in d['value_regions']], 'conv TF', round(d['roofline']['achieved'],1), 'loss', d['final_loss'])"
  done
done
