#!/usr/bin/env python3
"""Per-iteration cost and work counters of the cell-grid search on the bench workload (developer tool).
    python tools/grid_probe.py [points] [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402

capi = load_package().capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
group = int(sys.argv[3]) if len(sys.argv) > 3 else 5          # iterations per measurement, enqueued back to back (an idle GPU clocks down)
before, after = synth_cloud(np, n)
if os.environ.get("GRID_PROBE_SWAP") == "1":      # register the ROTATED cube onto the axis-aligned one: the fixed cloud then fills its grid's extent, and
    before, after = after, before                  # the moving cloud's shell hangs OUT of the extent (what an oriented grid would make of the bench workload)
with capi.Context(0) as ctx:
    ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=-1, sync_every=group))
    ctx.profile_enable(True)
    for it in range(0, iters, group):
        ctx.profile_reset()
        ctx.search_stats(True)
        ctx.icp_run(group)
        cand, rows, hard, pts, nodes, leaves, waves, cyc = ctx.search_stats(False)
        ms = {capi.KERNEL_NAMES[k]: ctx.profile_get(k) for k in range(len(capi.KERNEL_NAMES))}
        err = ctx.icp_result()[3]
        ms = {k: (v[0] / max(v[1], 1), v[1]) for k, v in ms.items()}
        print("it %2d+  nn %.3f ms  solve %.3f  flush %.3f+%.3f  err %.4g  cand/pt %.1f  rows/pt %.2f  hierarchy %.2f%%  waves %d  nodes/wave %.0f  leaves/wave %.0f  longest walk %d steps" % (
            it, ms["nn"][0], ms["solve"][0], ms["transform"][0], ms["finalize"][0], err, cand / max(pts, 1), rows / max(pts, 1),
            100.0 * hard / max(pts, 1), waves, nodes / max(waves, 1), leaves / max(waves, 1), cyc), flush=True)
