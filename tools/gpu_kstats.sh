#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command; prints the per-kernel summary and leaves it in gpurun_out/$1_kernel_stats.csv
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
name=${1:-kstats}
d=gpurun_out/prof_$name; rm -rf $d
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $d --output-format csv -- python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-sizes $BENCH_ARGS > $d.log 2>&1 || { tail -5 $d.log; exit 1; }
f=$(ls $d/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/${name}_kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %5s  avg %10.1f ns  total %6.2f %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), float(r["Percentage"])))
PY
tail -1 $d.log | cut -c1-400
