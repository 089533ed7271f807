#!/bin/bash
# round 3, GPU call 5: the full GPU suite + the default bench line + functional 2-rank bench over gloo on one GPU
mkdir -p gpurun_out/r3c5
O=gpurun_out/r3c5
timeout 2400 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest_full.log 2>&1
echo "pytest exit $?"; tail -25 $O/pytest_full.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cut -c1-1500 $O/bench.json; tail -3 $O/bench.err
ST_BENCH_SAME_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
   bench.py --gpus 2 --steps 5 --warmup 2 --size 1024 --dist-backend gloo > $O/bench_2rank_gloo.json 2> $O/bench_2rank.err
echo "2-rank gloo exit $?"; cut -c1-600 $O/bench_2rank_gloo.json; tail -3 $O/bench_2rank.err
