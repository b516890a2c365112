#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 6 gpurun_out/$name.log | cut -c1-700; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
step cpd_bench timeout -k 10 300 python tools/cpd_bench.py --big
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1; grep -i -E "mfma|VALU_BUSY|INSTS_VALU\b|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES" gpurun_out/counters_list.txt | head -40
rm -rf gpurun_out/prof_cpd_stats gpurun_out/prof_cpd_pmc
step prof_cpd_stats timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cpd_stats --output-format csv -- python3 tools/cpd_bench.py
step prof_cpd_pmc timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/prof_cpd_pmc --output-format csv -- python3 tools/cpd_bench.py
exit 0
