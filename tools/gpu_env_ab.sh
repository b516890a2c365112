#!/bin/bash
# A/B of environment settings on the bench workload: tools/gpu_env_ab.sh "VAR=a" "VAR=b OTHER=c" ...  (REPS rounds, interleaved)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in $(seq ${REPS:-1}); do
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-sizes --brute-ref-steps 0 ${BENCH_ARGS} 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('it/s %.0f  ms/step %.4f  nn avg %.4f ms  kernels %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], {k: round(v, 4) for k, v in d['kernels_ms_per_step'].items()}))
    else:
        print(l, end='')
" || exit 1
done
done
