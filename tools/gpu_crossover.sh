cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for n in 4000 6000 8000 10000 12000 16000; do
  for nn in brute grid; do
    echo -n "$n $nn: "
    timeout -k 10 300 python bench.py --points $n --nn $nn --steps 20 --warmup 5 --no-cpu-baseline --no-sizes --no-whole-call --brute-ref-steps 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('ms/step %.4f  nn avg %.4f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))
"
  done
done
