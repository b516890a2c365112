#!/bin/bash
# node visits per round (MISLAM_TREE_NODE_STEPS) for the wide walk, the static kernel and small moving clouds
mkdir -p gpurun_out
: > gpurun_out/steps2.log
b() { echo "$*" >> gpurun_out/steps2.log; env "$@" timeout -k 10 300 python bench.py --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'])" >> gpurun_out/steps2.log || exit 1; }
for k in 0 2 3 4 6; do b MISLAM_TREE_WIDE=1 MISLAM_TREE_NODE_STEPS=$k; done
for k in 0 4 6 8; do b MISLAM_TREE_DYNAMIC=0 MISLAM_TREE_NODE_STEPS=$k; done
EXTRA="--points 100000"
for k in 0 4 6 8; do b MISLAM_TREE_NODE_STEPS=$k; done
EXTRA="--points 10000000"
for k in 0 6; do b MISLAM_TREE_NODE_STEPS=$k; done
cat gpurun_out/steps2.log
for k in 0 6; do echo "k1t_vs_queries NODE_STEPS=$k"; MISLAM_TREE_NODE_STEPS=$k timeout -k 10 200 python tools/k1t_vs_queries.py; done
