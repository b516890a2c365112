#!/bin/bash
# HBM-side read traffic (FETCH_SIZE, KB per launch, uncorrected) of the hierarchy walk: one range per XCD, one global range, static.
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/parts_fetch.log
for cfg in "MISLAM_TREE_PARTS=8" "MISLAM_TREE_PARTS=1" "MISLAM_TREE_DYNAMIC=0"; do
  rm -rf gpurun_out/pf
  export $cfg
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pf --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 > gpurun_out/pf.log 2>&1 || { tail -5 gpurun_out/pf.log; exit 1; }
  unset MISLAM_TREE_PARTS MISLAM_TREE_DYNAMIC
  python3 - "$cfg" <<'PY' >> gpurun_out/parts_fetch.log
import csv, glob, sys
f = glob.glob("gpurun_out/pf/*/*_counter_collection.csv")[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "nn_tree_lane" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print(sys.argv[1], "launches", len(v), "FETCH_SIZE_KB mean", sum(v) / len(v))
PY
done
rm -rf gpurun_out/pf
cat gpurun_out/parts_fetch.log
