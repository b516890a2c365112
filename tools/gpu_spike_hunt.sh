#!/bin/bash
# developer tool: sporadic 20-60 ms stalls of a load / run -- how often, and in which host-side section (MISLAM_DEV_STALL_MS reports
# sections above the threshold with the thread's context switches).  Found: the container's CPU quota running out under numpy's
# BLAS pool (one spinning thread per visible core); with one BLAS thread (what bench.py, tests/ and tools/ now ask for) none are left.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export MISLAM_DEV_STALL_MS=8
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  OPENBLAS_NUM_THREADS=256 python tools/load_sweep.py performance > gpurun_out/spike_pool_$rep.log 2>&1
  python tools/load_sweep.py performance > gpurun_out/spike_quiet_$rep.log 2>&1
done
echo "BLAS pool on every visible core:"; grep -h "mislam stall" gpurun_out/spike_pool_*.log
echo "one BLAS thread:";   grep -h "mislam stall" gpurun_out/spike_quiet_*.log
cat /sys/fs/cgroup/cpu.max; grep -E "nr_periods|nr_throttled" /sys/fs/cgroup/cpu.stat
true
