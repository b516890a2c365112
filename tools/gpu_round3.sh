#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 3 gpurun_out/$name.log | cut -c1-900; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
step pytest_nn timeout -k 10 600 python -m pytest tests/test_gpu_nn.py tests/test_gpu_icp.py -m gpu -q -s --timeout 300 -x
step bench_1e6 timeout -k 10 300 python bench.py --no-cpu-baseline --brute-ref-steps 0
step bench_1e6_late timeout -k 10 300 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup 60 --steps 10
MISLAM_TREE_R=1 step bench_1e6_late_wave timeout -k 10 300 python bench.py --no-cpu-baseline --brute-ref-steps 0 --warmup 60 --steps 10
step bench_1e5 timeout -k 10 300 python bench.py --points 100000 --steps 50 --warmup 5 --no-cpu-baseline --brute-ref-steps 0
step bench_15k_tree timeout -k 10 300 python bench.py --points 14904 --steps 200 --warmup 10 --no-cpu-baseline --nn tree --brute-ref-steps 0
step bench_1e7 timeout -k 10 300 python bench.py --points 10000000 --steps 10 --warmup 2 --no-cpu-baseline --brute-ref-steps 0
exit 0
