R=$GRAFT_REPO_ROOT; cd $R
run() { python bench.py --no-extra --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms')"; }
for rep in 1 2; do
  echo "L0 default:   $(ST_STREAM_LAYOUT=0 run --steps 20 --warmup 5)"
  echo "L1 default:   $(ST_STREAM_LAYOUT=1 run --steps 20 --warmup 5)"
  echo "L1 nograph:   $(ST_STREAM_LAYOUT=1 ST_HEAD_GRAPH=0 run --steps 20 --warmup 5)"
  echo "L0 nograph:   $(ST_STREAM_LAYOUT=0 ST_HEAD_GRAPH=0 run --steps 20 --warmup 5)"
done
echo "L0 long:      $(ST_STREAM_LAYOUT=0 run --steps 300 --warmup 100)"
echo "L1 long:      $(ST_STREAM_LAYOUT=1 run --steps 300 --warmup 100)"
echo "L0 nograph long: $(ST_STREAM_LAYOUT=0 ST_HEAD_GRAPH=0 run --steps 300 --warmup 100)"
echo "L1 nograph long: $(ST_STREAM_LAYOUT=1 ST_HEAD_GRAPH=0 run --steps 300 --warmup 100)"
