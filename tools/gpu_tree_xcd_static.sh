#!/bin/bash
# static walk, 6 node visits per round: XCD-contiguous chunk mapping on / off -- time (1e6, 1e7, 1e5) and FETCH_SIZE (1e6)
mkdir -p gpurun_out; export TMPDIR=/tmp
export MISLAM_TREE_DYNAMIC=0 MISLAM_TREE_NODE_STEPS=6
: > gpurun_out/xcd_static.log
for x in 1 0; do
  export MISLAM_TREE_XCD_CHUNKS=$x
  for pts in 1000000 10000000 100000; do
    echo "xcd_chunks=$x points=$pts" >> gpurun_out/xcd_static.log
    timeout -k 10 300 python bench.py --points $pts --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['kernel'], d['config']['error_after_steps'])" >> gpurun_out/xcd_static.log || exit 1
  done
  rm -rf gpurun_out/pf
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pf --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 > gpurun_out/pf.log 2>&1 || { tail -5 gpurun_out/pf.log; exit 1; }
  python3 - <<'PY' >> gpurun_out/xcd_static.log
import csv, glob
f = glob.glob("gpurun_out/pf/*/*_counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "nn_tree_lane" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("  FETCH_SIZE_KB mean per launch", sum(v) / len(v), "launches", len(v))
PY
done
rm -rf gpurun_out/pf
cat gpurun_out/xcd_static.log
timeout -k 10 300 python -m pytest tests/test_gpu_nn.py -x -q -m gpu 2>&1 | tail -2
