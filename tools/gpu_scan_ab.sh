#!/bin/bash
# parity of the grid scan (NN + ICP + soak tests), A/B of the product build against variant builds (args), per-iteration probe and wave timeline
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${TAG:-s}
timeout -k 10 900 python -m pytest tests/test_gpu_nn.py tests/test_gpu_icp.py tests/test_gpu_nn_soak.py -x -q -m gpu > gpurun_out/r04_${tag}_tests.log 2>&1 || { tail -30 gpurun_out/r04_${tag}_tests.log; exit 1; }
tail -2 gpurun_out/r04_${tag}_tests.log
REPS=${REPS:-2} bash tools/gpu_ab.sh "$@" > gpurun_out/r04_${tag}_ab.log 2>&1 || { tail -20 gpurun_out/r04_${tag}_ab.log; exit 1; }
cat gpurun_out/r04_${tag}_ab.log
unset MISLAM_LIB
timeout -k 10 300 python tools/grid_probe.py 1000000 30 > gpurun_out/r04_${tag}_probe.log 2>&1 || exit 1
cat gpurun_out/r04_${tag}_probe.log
if [ -f cuda-slam_amd/variants/libmislam_timeline.so ]; then
  MISLAM_LIB=$GRAFT_REPO_ROOT/cuda-slam_amd/variants/libmislam_timeline.so timeout -k 10 300 python tools/wave_timeline.py 6 12 20 > gpurun_out/r04_${tag}_timeline.log 2>&1
  cat gpurun_out/r04_${tag}_timeline.log
fi
