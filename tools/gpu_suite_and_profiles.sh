#!/bin/bash
# Full GPU suite, rocprofv3 kernel stats and HBM counters of the default bench command, then the driver's default bench
# (which reads the fresh counter summary for roofline.traffic).  Copy gpurun_out/{bench_default.log,bench_kernel_stats.csv,
# hbm_counters.json} into profiles/ afterwards.
mkdir -p gpurun_out
export TMPDIR=/tmp
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 3 gpurun_out/$name.log | cut -c1-600; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
step pytest_gpu timeout -k 10 900 python -m pytest tests -m gpu -q -s --timeout 600
step smoke timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()"
rm -rf gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
step prof_stats timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 2
step prof_fetch timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 1
step prof_write timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write --output-format csv -- python3 bench.py --steps 50 --warmup 2 --no-cpu-baseline --brute-ref-steps 1
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/hbm_counters.json
cp gpurun_out/hbm_counters.json profiles/r01_bench_n1e6_hbm_counters.json      # on the box: what the bench below reads
f=$(find gpurun_out/prof_stats -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/bench_kernel_stats.csv
find gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write -type f ! -name '*stats*' -delete
step bench_default timeout -k 10 300 python bench.py
exit 0
