#!/bin/bash
# A/B of the dynamically fetching hierarchy walk: parity suite with it forced on, then timings per refill threshold.
mkdir -p gpurun_out
export MISLAM_TREE_DYNAMIC=1
timeout -k 10 300 python -m pytest tests/test_gpu_nn.py -x -q -m gpu > gpurun_out/dyn_parity.log 2>&1 || { tail -20 gpurun_out/dyn_parity.log; exit 1; }
tail -2 gpurun_out/dyn_parity.log
: > gpurun_out/dyn_times.log
for r in ${REFILLS:-64 32 16 8 1}; do
  echo "REFILL=$r" >> gpurun_out/dyn_times.log
  MISLAM_TREE_REFILL=$r timeout -k 10 200 python tools/k1t_vs_queries.py >> gpurun_out/dyn_times.log 2>&1 || { tail -5 gpurun_out/dyn_times.log; exit 1; }
done
echo "STATIC" >> gpurun_out/dyn_times.log
MISLAM_TREE_DYNAMIC=0 timeout -k 10 200 python tools/k1t_vs_queries.py >> gpurun_out/dyn_times.log 2>&1
cat gpurun_out/dyn_times.log
