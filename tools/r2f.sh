R=$GRAFT_REPO_ROOT; cd $R
for sz in 128 256 512; do echo "== $sz"; ST_AMD_TIMELINE=1 timeout 100 python bench.py --size $sz --steps 40 --warmup 10 --no-extra --no-cpu-baseline 2>&1 | grep timeline | tail -2; done
