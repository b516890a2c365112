#!/bin/bash
# round 3, step 1: parity of the rewritten grid scan, then A/B against round 2's scan (variant v1scan) on the bench workload
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_icp.py tests/test_gpu_nn_soak.py -x -q -m gpu > gpurun_out/r03_s1_tests.log 2>&1 || { tail -30 gpurun_out/r03_s1_tests.log; exit 1; }
tail -3 gpurun_out/r03_s1_tests.log
REPS=2 bash tools/gpu_ab.sh v1scan > gpurun_out/r03_s1_ab.log 2>&1 || { tail -20 gpurun_out/r03_s1_ab.log; exit 1; }
cat gpurun_out/r03_s1_ab.log
for v in product v1scan; do
  if [ $v = product ]; then unset MISLAM_LIB; else export MISLAM_LIB=$GRAFT_REPO_ROOT/cuda-slam_amd/variants/libmislam_$v.so; fi
  echo "== $v" >> gpurun_out/r03_s1_probe.log
  timeout -k 10 300 python tools/grid_probe.py 1000000 30 >> gpurun_out/r03_s1_probe.log 2>&1 || exit 1
done
cat gpurun_out/r03_s1_probe.log
