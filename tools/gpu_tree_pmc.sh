#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/pmc_tree1 gpurun_out/pmc_tree2
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d gpurun_out/pmc_tree1 --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 > gpurun_out/pmc_tree1.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM -d gpurun_out/pmc_tree2 --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --brute-ref-steps 0 > gpurun_out/pmc_tree2.log 2>&1
python3 - <<'PY'
import csv,glob,collections
out={}
kernel_name="nn_tree_lane"
for d in ("pmc_tree1","pmc_tree2"):
    fs=glob.glob(f"gpurun_out/{d}/*/*_counter_collection.csv")
    if not fs: print(d,"no csv"); continue
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "nn_tree_lane" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            kernel_name = r["Kernel_Name"].split("<")[0].split("::")[-1].split("(")[0]
    for k,v in agg.items(): print(d,k,len(v),sum(v)/len(v))
    out.update({k: sum(v)/len(v) for k,v in agg.items()})
import json
json.dump({"command": "rocprofv3 --pmc <two counter sets, separate passes> -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --brute-ref-steps 0",
           "kernel": kernel_name, "per_launch_mean": out}, open("gpurun_out/tree_counters.json","w"), indent=1)
PY
tail -2 gpurun_out/pmc_tree2.log | cut -c1-300
