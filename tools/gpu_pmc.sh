#!/bin/bash
# rocprofv3 --pmc passes over the bench command, per-launch means for kernels whose name contains $KERNEL (default nn_grid_kernel).
#   tools/gpu_pmc.sh OUT.json "CTR1 CTR2 ..." ["CTR ..." more passes]      (each quoted list = one pass; runs on the GPU box)
# Env: STEPS, WARMUP, BENCH_ARGS, KERNEL, MISLAM_LIB
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=$1; shift
k=${KERNEL:-nn_grid_kernel}
i=0
dirs=""
for set in "$@"; do
  i=$((i+1)); d=gpurun_out/pmc_pass$i; rm -rf $d
  timeout -k 10 300 rocprofv3 --pmc $set -d $d --output-format csv -- python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-sizes --brute-ref-steps 0 $BENCH_ARGS > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
  dirs="$dirs $d"
done
python3 - $out $k "$*" $dirs <<'PY'
import csv, glob, collections, json, sys
out, kern, sets = sys.argv[1], sys.argv[2], sys.argv[3]
res = {}
launches = 0
for d in sys.argv[4:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in agg.items():
        res[c] = sum(v) / len(v)
        launches = len(v)
json.dump({"kernel": kern, "launches_per_pass": launches, "command": "rocprofv3 --pmc <one pass per counter set: %s> -- python3 bench.py --steps N --warmup W --no-cpu-baseline --no-sizes --brute-ref-steps 0" % sets,
           "per_launch_mean": res}, open(out, "w"), indent=1)
for c in sorted(res): print("%-32s %.4g" % (c, res[c]))
PY
