#!/bin/bash
# A/B of the two per-lane walks (MISLAM_TREE_COMPACT=0: the 64-byte node records and float4 leaf points) on the bench workload,
# early and late ICP iterations, then the NN / ICP parity suites under the default.
mkdir -p gpurun_out
for c in 0 1; do
  export MISLAM_TREE_COMPACT=$c
  for w in 2 60; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup $w > gpurun_out/b_ab.log 2>&1 || { tail -5 gpurun_out/b_ab.log; exit 1; }
    python -c "
import json;d=json.loads(open('gpurun_out/b_ab.log').read().strip().splitlines()[-1]);print('compact=$c warmup=$w', round(d['value'],1), 'it/s', round(d['ms_per_step'],3), 'ms/step', round(d['roofline']['avg_launch_ms'],3), 'ms/search')"
  done
done
unset MISLAM_TREE_COMPACT
timeout -k 10 800 python -m pytest tests/test_gpu_nn.py tests/test_gpu_icp.py tests/test_gpu_nicp.py -q -m gpu 2>&1 | tail -3
