#!/bin/bash
# Profiles (round tag R, default r06) of the driver's default bench command (python bench.py --steps 20 --warmup 5): rocprofv3 kernel-trace stats + the
# rows of the TIMED dispatches, the --pmc passes (one per counter set, never mixed with tracing) for the search kernel, the every-pair
# kernel and the CPD E-step kernels of the same run, and the same SQ counters on tools/valu_probe (kernels of a known instruction
# count at 8 waves per SIMD) -- what bench.py's `issue` rooflines are calibrated by.
#   gpurun -- 'bash tools/gpu_profiles.sh'   ->   gpurun_out/<R>_*  (copy the summaries into profiles/)
mkdir -p gpurun_out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
S=${STEPS:-20}; W=${WARMUP:-5}; R=${R:-r06}; export R
export MISLAM_BENCH_NO_RCCL_FLOOR=1   # (the one-rank RCCL communicator of the all-reduce floor leg is not part of what is profiled)
# The shipped default is profiled (cooperative K-centre sweep ON).  ROCm 7.2's HIP runtime faults in its own static destructor when a process that made
# ANY cooperative launch ends under rocprofv3 -- after the tool has written its files (profiles/r06_coop_exit_abort.md: reproduced by a 50-line program
# without libmislam, frames symbolised) -- so a pass may end with exit code 139 if and only if its log shows the tool's finalisation before the abort.
B="python3 bench.py --steps $S --warmup $W --no-cpu-baseline --no-sizes --no-whole-call"
run() { d=gpurun_out/${R}_$1; shift; rm -rf $d; timeout -k 10 500 rocprofv3 "$@" -d $d --output-format csv -- $B > $d.log 2>&1; rc=$?
        if [ $rc -eq 139 ] && grep -q "tool finalization" $d.log && grep -q '"metric"' $d.log; then echo "pass $d done (exit 139 AFTER the tool's finalisation: the runtime's cooperative-queue teardown, r06_coop_exit_abort.md)"
        elif [ $rc -ne 0 ]; then tail -5 $d.log; exit 1; else echo "pass $d done"; fi; }
probe() { d=gpurun_out/${R}_$1; shift; rm -rf $d; timeout -k 10 200 rocprofv3 "$@" -d $d --output-format csv -- tools/valu_probe > $d.log 2>&1 || { tail -5 $d.log; exit 1; }; }
[ -x tools/valu_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -w tools/valu_probe.hip -o tools/valu_probe || exit 1
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"
SQ2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run sq1 --pmc $SQ1
run sq2 --pmc $SQ2
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
# the matrix pipe under the CPD contraction (north_star: "MFMA utilisation against the chip's peak"): busy cycles and fp32 MFMA operations
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
probe probe_trace --kernel-trace
probe probe_sq1 --pmc $SQ1
probe probe_sq2 --pmc $SQ2
python3 tools/profiles_reduce.py $S $W || exit 1
find gpurun_out/${R}_stats gpurun_out/${R}_fetch gpurun_out/${R}_write gpurun_out/${R}_sq1 gpurun_out/${R}_sq2 gpurun_out/${R}_tcp gpurun_out/${R}_mfma gpurun_out/${R}_probe_trace gpurun_out/${R}_probe_sq1 gpurun_out/${R}_probe_sq2 -type f -delete 2>/dev/null
exit 0
