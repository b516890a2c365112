#!/bin/bash
mkdir -p gpurun_out
for L in 4 8 16 32; do
  if [ $L -eq 8 ]; then unset MISLAM_LIB; else export MISLAM_LIB=$PWD/tools/libmislam_leaf$L.so; fi
  for W in 2 60; do
    echo "leaf=$L warmup=$W $(timeout -k 10 120 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup $W --steps 10 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' ')"
  done
  echo "leaf=$L 1e7 $(timeout -k 10 200 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup 2 --steps 5 --points 10000000 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"avg_launch_ms": [0-9.]*' | tr '\n' ' ')"
done
