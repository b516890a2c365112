#!/bin/bash
# bench.py at the sizes DESIGN.md's measurement table quotes (default path; the every-pair column comes from --nn brute)
mkdir -p gpurun_out
run() { label=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --brute-ref-steps 0 "$@" > gpurun_out/b_sz.log 2>&1 || { tail -3 gpurun_out/b_sz.log; return; }
  python -c "
import json;d=json.loads(open('gpurun_out/b_sz.log').read().strip().splitlines()[-1]);print('$label', round(d['value'],2), 'it/s', round(d['ms_per_step'],4), 'ms/step', round(d['roofline']['avg_launch_ms'],4), 'ms/search', d['config']['nn'])"; }
run "1e6 steps5-55" --warmup 5 --steps 50
run "1e7" --points 10000000 --steps 5
run "1e5" --points 100000 --steps 20
run "1e5 brute" --points 100000 --steps 10 --nn brute
run "14904 tree" --points 14904 --steps 50 --nn tree
run "14904 auto" --points 14904 --steps 50
