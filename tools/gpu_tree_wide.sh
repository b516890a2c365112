#!/bin/bash
# The 4-wide hierarchy walk: NN parity suite with it forced on, then bench.py at 1e6 (and 1e7) against the default walk.
mkdir -p gpurun_out
MISLAM_TREE_WIDE=1 timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -x -q -m gpu > gpurun_out/wide_parity.log 2>&1 || { tail -30 gpurun_out/wide_parity.log; exit 1; }
tail -2 gpurun_out/wide_parity.log
: > gpurun_out/wide_bench.log
for n in ${SIZES:-1000000 10000000}; do
  for cfg in "MISLAM_TREE_WIDE=1" "MISLAM_TREE_WIDE=0"; do
    echo "points=$n $cfg" >> gpurun_out/wide_bench.log
    env $cfg timeout -k 10 300 python bench.py --points $n --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['config']['error_after_steps'])" >> gpurun_out/wide_bench.log || exit 1
  done
done
cat gpurun_out/wide_bench.log
