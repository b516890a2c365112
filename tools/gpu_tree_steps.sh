#!/bin/bash
# node visits per round of the dynamically fetching walk (MISLAM_TREE_NODE_STEPS): parity, then bench.py at 1e6 per setting
mkdir -p gpurun_out
: > gpurun_out/steps_bench.log
for k in ${STEPS:-0 1 2 3 4 6 8}; do
  export MISLAM_TREE_NODE_STEPS=$k
  if [ "$k" = "2" ]; then timeout -k 10 300 python -m pytest tests/test_gpu_nn.py -x -q -m gpu > gpurun_out/steps_parity.log 2>&1 || { tail -20 gpurun_out/steps_parity.log; exit 1; }; fi
  echo "node_steps=$k" >> gpurun_out/steps_bench.log
  timeout -k 10 300 python bench.py --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['error_after_steps'])" >> gpurun_out/steps_bench.log || exit 1
done
cat gpurun_out/steps_bench.log
