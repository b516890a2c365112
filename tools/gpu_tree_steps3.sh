#!/bin/bash
# with bounded node visits per round: static vs dynamically fetching vs wide walk at 1e6 and 1e7 points
mkdir -p gpurun_out
: > gpurun_out/steps3.log
b() { echo "$EXTRA $*" >> gpurun_out/steps3.log; env "$@" timeout -k 10 300 python bench.py --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'])" >> gpurun_out/steps3.log || exit 1; }
for EXTRA in "--points 1000000" "--points 10000000" "--points 3000000"; do
  b MISLAM_TREE_DYNAMIC=0 MISLAM_TREE_NODE_STEPS=6
  b MISLAM_TREE_DYNAMIC=0 MISLAM_TREE_NODE_STEPS=5
  b MISLAM_TREE_DYNAMIC=1 MISLAM_TREE_NODE_STEPS=6
  b MISLAM_TREE_DYNAMIC=1 MISLAM_TREE_NODE_STEPS=6 MISLAM_TREE_REFILL=32
  b MISLAM_TREE_DYNAMIC=1 MISLAM_TREE_NODE_STEPS=6 MISLAM_TREE_REFILL=16
  b MISLAM_TREE_WIDE=1 MISLAM_TREE_NODE_STEPS=3
done
cat gpurun_out/steps3.log
