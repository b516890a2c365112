#!/bin/bash
mkdir -p gpurun_out
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 3 gpurun_out/$name.log | cut -c1-300; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
step pytest_nn timeout -k 10 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_icp.py -m gpu -q --timeout 300 -x
for R in 0 -1; do
  export MISLAM_TREE_R=$R
  for W in 2 60; do
    echo "R=$R warmup=$W $(timeout -k 10 120 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup $W --steps 10 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' ')"
  done
  echo "R=$R 1e7 $(timeout -k 10 200 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup 2 --steps 5 --points 10000000 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' ')"
  echo "R=$R 1e5 $(timeout -k 10 200 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup 5 --steps 50 --points 100000 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' ')"
done
