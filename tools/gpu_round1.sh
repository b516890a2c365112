#!/bin/bash
# One gpurun call: GPU parity suite, smoke, bench, torch-first runtime check.  A step that was killed (124/137) ends the call.
mkdir -p gpurun_out
step() { name=$1; shift; echo "== $name"; "$@" > gpurun_out/$name.log 2>&1; rc=$?; echo "$name rc=$rc"; tail -n 6 gpurun_out/$name.log; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed: stopping"; exit $rc; fi; }
step pytest_gpu timeout -k 10 800 python -m pytest tests -m gpu -q -x --timeout 600
step smoke timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()"
step bench timeout -k 10 300 python bench.py
step torch_first timeout -k 10 200 python -c "import torch; import __graft_entry__ as g; g.smoke(); print('torch', torch.__version__, torch.version.hip)"
exit 0
