#!/bin/bash
# Average share of a wave's 64 lanes that are active per VALU instruction of the hierarchy walk (SQ_THREAD_CYCLES_VALU /
# SQ_ACTIVE_INST_VALU / 64), dynamic and static forms.
mkdir -p gpurun_out; export TMPDIR=/tmp
: > gpurun_out/lane_util.log
for cfg in ${CONFIGS:-MISLAM_TREE_DYNAMIC=1 MISLAM_TREE_DYNAMIC=0}; do
  rm -rf gpurun_out/lu
  export $cfg
  timeout -k 10 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS -d gpurun_out/lu --output-format csv -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 > gpurun_out/lu.log 2>&1 || { tail -5 gpurun_out/lu.log; exit 1; }
  python3 - "$cfg" <<'PY' >> gpurun_out/lane_util.log
import csv, glob, sys, collections
f = glob.glob("gpurun_out/lu/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "nn_tree_lane" in r["Kernel_Name"] or "nn_tree_wide" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
print(sys.argv[1], {k: round(v) for k, v in m.items()})
if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_ACTIVE_INST_VALU"):
    print("  active lanes per VALU instruction: %.1f of 64" % (m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]))
PY
done
rm -rf gpurun_out/lu
cat gpurun_out/lane_util.log
