#!/bin/bash
# The NN parity suite, then bench.py at 1e6 and 1e7 points with the dynamically fetching walk: one range per XCD (default), one
# global range, and the static kernel.
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_nn.py -x -q -m gpu > gpurun_out/dyn_suites.log 2>&1 || { tail -30 gpurun_out/dyn_suites.log; exit 1; }
tail -2 gpurun_out/dyn_suites.log
: > gpurun_out/dyn_bench.log
for n in 1000000 10000000; do
  for cfg in "MISLAM_TREE_PARTS=8" "MISLAM_TREE_PARTS=1" "MISLAM_TREE_DYNAMIC=0"; do
    echo "points=$n $cfg" >> gpurun_out/dyn_bench.log
    env $cfg timeout -k 10 300 python bench.py --points $n --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'])" >> gpurun_out/dyn_bench.log || exit 1
  done
done
cat gpurun_out/dyn_bench.log
