#!/bin/bash
# round 3: counter passes (SQ + TA/TCP/TD) over the bench command for the product build and the variants named on the command line
for v in product "$@"; do
  if [ $v = product ]; then unset MISLAM_LIB; else export MISLAM_LIB=$GRAFT_REPO_ROOT/cuda-slam_amd/variants/libmislam_$v.so; fi
  echo "== $v"
  bash tools/gpu_pmc.sh gpurun_out/r03_pmc_$v.json \
    "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
    "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
    "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" || exit 1
done
