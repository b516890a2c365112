cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in product noitems product noitems; do
  if [ $v = product ]; then unset MISLAM_LIB; else export MISLAM_LIB=$GRAFT_REPO_ROOT/cuda-slam_amd/variants/libmislam_$v.so; fi
  for n in 100000 10000000; do
    echo "== $v $n"
    timeout -k 10 300 python bench.py --points $n --steps 10 --warmup 5 --no-cpu-baseline --no-sizes --no-whole-call --brute-ref-steps 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('it/s %.0f  ms/step %.4f  nn avg %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))
"
  done
done
