#!/bin/bash
# Developer: the headline bench (search legs only) under values of one environment knob.   tools/knob_sweep.sh MISLAM_GRID_PPC 1.0 1.25 1.5
knob=$1; shift
for v in "$@"; do for i in 1 2; do
  env $knob=$v python bench.py --no-cpd --no-whole-call --no-cpu-baseline --no-sizes 2>/dev/null > /tmp/_knob.json
  python - "$knob" "$v" <<'PY'
import json, sys
for ln in open('/tmp/_knob.json'):
    if ln.startswith('{'):
        d = json.loads(ln); print(sys.argv[1], sys.argv[2], round(d['value']), round(d['roofline']['avg_launch_ms'], 5), flush=True)
PY
done; done
