#!/bin/bash
# Full GPU suite (no -x), benches at three sizes, rocprofv3 kernel stats + HBM counters for the bench.
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 4 gpurun_out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
step pytest_gpu timeout -k 10 900 python -m pytest tests -m gpu -q -s --timeout 600
step bench_1e6 timeout -k 10 300 python bench.py
step bench_1e5 timeout -k 10 300 python bench.py --points 100000 --steps 50 --warmup 5 --no-cpu-baseline
step bench_15k timeout -k 10 300 python bench.py --points 14904 --steps 200 --warmup 10 --no-cpu-baseline
step prof_stats timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
step prof_fetch timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
step prof_write timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
find gpurun_out -name "*.csv" | head -30
exit 0
