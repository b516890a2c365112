#!/bin/bash
# static walk: runs of S consecutive 256-point chunks per XCD (MISLAM_TREE_XCD_CHUNKS=S; 0 = plain mapping, 1 = one contiguous
# eighth per XCD) -- search time at 1e6 / 1e7 and FETCH_SIZE at 1e6
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/xcd_runs.log
for x in ${RUNS:-0 1 4 8 16 32 64 128}; do
  export MISLAM_TREE_XCD_CHUNKS=$x
  for pts in 1000000 10000000; do
    echo -n "xcd_chunks=$x points=$pts: " >> gpurun_out/xcd_runs.log
    timeout -k 10 300 python bench.py --points $pts --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value'],1), round(d['roofline']['avg_launch_ms'],4), d['config']['error_after_steps'])" >> gpurun_out/xcd_runs.log || exit 1
  done
  rm -rf gpurun_out/pf
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pf --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 > gpurun_out/pf.log 2>&1 || { tail -5 gpurun_out/pf.log; exit 1; }
  python3 - <<'PY' >> gpurun_out/xcd_runs.log
import csv, glob
f = glob.glob("gpurun_out/pf/*/*_counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "nn_tree_lane" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("  FETCH_SIZE_KB mean per launch", round(sum(v) / len(v)))
PY
done
rm -rf gpurun_out/pf
cat gpurun_out/xcd_runs.log
