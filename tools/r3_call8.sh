#!/bin/bash
mkdir -p gpurun_out/r3c8
O=gpurun_out/r3c8
timeout 600 python -m pytest tests/test_stylize_sharded_gpu.py::test_stylize_lbfgs_in_separate_processes_matches_single_gpu "tests/test_hot_path_gpu.py::test_closure_against_reference_goldens" -q -s > $O/pytest.log 2>&1
echo "pytest exit $?"; tail -4 $O/pytest.log; grep "stylize-sharded" $O/pytest.log
for size in 2048 1024; do
 for spare in 0 8 16 32; do
  for rep in 1 2; do
   v=$(ST_CONV_PC_SPARE=$spare timeout 200 python bench.py --no-extra --no-cpu-baseline --size $size --steps 12 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s conv %.0f TF' % (d['value'], d['roofline']['achieved']))")
   echo "[spare] size $size spare $spare: $v"
  done
 done
done
for spare in 0 8; do
  v=$(ST_CONV_PC_SPARE=$spare timeout 200 python bench.py --no-extra --no-cpu-baseline --size 512 --steps 60 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.2f it/s' % d['value'])")
  echo "[spare] size 512 spare $spare: $v"
done
