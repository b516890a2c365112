#!/bin/bash
# bench under a list of environment settings: tools/gpu_env_sweep.sh "A=1" "A=2 B=3" ...   (developer tool, runs on the GPU box)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-sizes --brute-ref-steps 0 ${BENCH_ARGS} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('it/s %.0f  ms/step %.4f  nn avg %.4f ms  kernels %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v, 4) for k, v in d['kernels_ms_per_step'].items()}))
" || exit 1
done
