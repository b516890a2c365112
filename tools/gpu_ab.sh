#!/bin/bash
# A/B of variant builds on the bench workload: [REPS=3] tools/gpu_ab.sh NAME... (product build first; REPS rounds, interleaved:
# one bench run differs from the next by a few per cent)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in $(seq ${REPS:-1}); do
for v in product "$@"; do
  if [ $v = product ]; then unset MISLAM_LIB; else export MISLAM_LIB=$GRAFT_REPO_ROOT/cuda-slam_amd/variants/libmislam_$v.so; fi
  echo "== $v"
  timeout -k 10 300 python bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} --no-cpu-baseline --no-sizes --no-cpd --no-whole-call --brute-ref-steps 0 ${BENCH_ARGS} 2>&1 | python -c "
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
