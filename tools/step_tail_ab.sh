# A/B of the iteration's tail: fold + total + update + bound clearing in ONE launch (ST_STEP_TAIL=2, shipped), total + clearing in
# the update kernel (1), four launches (0); separate processes, interleaved.  Needs an --experiments library.   gpurun -- bash tools/step_tail_ab.sh "128 256 512"
R=$GRAFT_REPO_ROOT
for SZ in ${1:-128 512}; do
  for V in 0 1 2 0 1 2; do
    echo "== ST_STEP_TAIL=$V bench $SZ"; ST_STEP_TAIL=$V python $R/bench.py --no-extra --no-cpu-baseline --no-pmc --size $SZ --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  it/s', round(d['value'],2), 'regions', [round(x,2) for x in d['value_regions']], 'loss', d['final_loss'])"
  done
done
