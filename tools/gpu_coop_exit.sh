#!/bin/bash
# VERDICT r05 item 4 / ADVICE r05 (medium): who faults at exit when a process that made a cooperative launch ran under rocprofv3?
#   gpurun -- 'bash tools/gpu_coop_exit.sh'  ->  gpurun_out/coop_exit/*.log, *.maps, summary.txt
# Every case is a fresh process; each leaves its exit code and (on a fault) the abort's raw frames next to the process's own map of libraries.
mkdir -p gpurun_out/coop_exit; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/coop_exit
[ -x tools/coop_exit_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/coop_exit_probe.hip -o tools/coop_exit_probe || exit 1
: > $O/summary.txt
for mode in plain coop coop_stream coop_reset; do
  timeout -k 10 120 tools/coop_exit_probe $mode $O/bare_$mode.maps > $O/bare_$mode.log 2>&1; echo "bare $mode: exit $?" >> $O/summary.txt
  rm -rf $O/prof_$mode
  timeout -k 10 180 rocprofv3 --kernel-trace --stats -d $O/prof_$mode --output-format csv -- tools/coop_exit_probe $mode $O/prof_$mode.maps > $O/prof_$mode.log 2>&1; echo "rocprofv3 $mode: exit $?" >> $O/summary.txt
done
# the library's own cooperative kernel (K-centre sweep of 49 000 points), context closed explicitly before the interpreter ends
cat > $O/_kc.py <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from __graft_entry__ import load_package
capi = load_package().capi
x = np.random.default_rng(1).uniform(-1, 1, (49000, 3)).astype(np.float32)
with capi.Context(0) as ctx:
    c, l = ctx.fgt_kcenter(x, 51)
    print("labels", int(l.max()) + 1)
open(sys.argv[1], "w").writelines(ln for ln in open("/proc/self/maps") if ".so" in ln and " r-xp " in ln)
print("closed")
PY
timeout -k 10 300 python3 $O/_kc.py $O/bare_mislam.maps > $O/bare_mislam.log 2>&1; echo "bare libmislam K-centre 49k: exit $?" >> $O/summary.txt
rm -rf $O/prof_mislam
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_mislam --output-format csv -- python3 $O/_kc.py $O/prof_mislam.maps > $O/prof_mislam.log 2>&1; echo "rocprofv3 libmislam K-centre 49k: exit $?" >> $O/summary.txt
grep -h "coop" $O/prof_mislam/*/*kernel_stats.csv > $O/prof_mislam_coop_rows.csv 2>/dev/null
find $O -name "*.csv" -path "*prof_*/*" ! -name "*kernel_stats.csv" -delete 2>/dev/null
cat $O/summary.txt
exit 0
