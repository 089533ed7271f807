OUT=gpurun_out/r5s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_sharding_gpu.py tests/test_large_strips_gpu.py tests/test_sharding_multiproc_gpu.py tests/test_stylize_sharded_gpu.py -q -x -n 4 2>&1 | tail -5 > $OUT/tests.txt
for b in 0 1 0 1; do for cfg in "2896x2172 8" "2048 4"; do echo "ST_STRIP_HALO_BOUND=$b $cfg"; ST_STRIP_HALO_BOUND=$b timeout 300 python tools/strip_bench.py $cfg 2>&1 | grep -E "per-rank|rank 7|rank 3"; done; done > $OUT/strip_ab.txt 2>&1
