#!/bin/bash
# every-pair vs cell-grid search at small sizes (where MI_NN_AUTO should switch): ms per ICP step and per search
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in ${SIZES:-4000 8000 12000 16000 24000 32000 50000}; do
  for nn in brute grid; do
    timeout -k 10 200 python bench.py --points $n --nn $nn --steps 20 --warmup 5 --no-cpu-baseline --no-sizes --brute-ref-steps 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('n %7d  %-5s  ms/step %.4f  nn avg %.4f ms' % ($n, '$nn', d['ms_per_step'], d['roofline']['avg_launch_ms']))
" || exit 1
  done
done
