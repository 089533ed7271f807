# A/B of the fat convolution kernel's selection rule in the iteration: ST_CONV_FAT_ROUNDS = 2 (shipped: at least two whole
# rounds of workgroups) against 1 (one whole round is enough), separate processes, interleaved.  Needs an --experiments library
# (the default build does not read these switches from the environment).   gpurun -- bash tools/fat_rounds_ab.sh "1024 1448"
R=$GRAFT_REPO_ROOT
for SZ in ${1:-1024}; do
  for V in 2 1 2 1; do
    echo "== ST_CONV_FAT_ROUNDS=$V bench $SZ"; ST_CONV_FAT_ROUNDS=$V python $R/bench.py --no-extra --no-cpu-baseline --no-pmc --size $SZ --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  it/s', round(d['value'],2), 'regions', [round(x,2) for x in d['value_regions']], 'conv TF', round(d['roofline']['achieved'],1), 'loss', d['final_loss'])"
  done
done
