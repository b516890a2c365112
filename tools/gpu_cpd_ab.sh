#!/bin/bash
# A/B of variant builds on the CPD bench (tools/cpd_bench.py --big): K7a / K7b per launch, VALU and MFMA forms
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in product "$@"; do
  if [ $v = product ]; then unset MISLAM_LIB; else export MISLAM_LIB=$GRAFT_REPO_ROOT/cuda-slam_amd/variants/libmislam_$v.so; fi
  echo "== $v"
  timeout -k 10 300 python tools/cpd_bench.py --big 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        if 'contraction' in d: print('%-16s %-10s it %d  total %.2f ms  K7a %.4f  K7b %.4f ms' % (d['case'], d['contraction'], d['iterations'], d['wall_ms_total'], d['K7a_denominator_ms'], d['K7b_contraction_ms']))
    else: print(l, end='')
"
done
