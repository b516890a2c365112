#!/usr/bin/env python3
"""Dynamic side of the search kernel's instruction budget: the loop trip counts of the TIMED launches of the default bench
command (iterations warmup+1 .. warmup+steps of the 1e6-point registration), from the counting build of the same kernel
(mi_profile_search_stats + mi_profile_search_phases: same control flow, extra counters).  tools/isa_budget.py --compose multiplies the
warm kernel's static per-phase instruction counts by them.
    python tools/phase_counts.py [steps] [warmup] [points]  ->  one JSON line"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402

capi = load_package().capi
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1000000
before, after = synth_cloud(np, n)
with capi.Context(0) as ctx:
    ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=-1, sync_every=max(steps, 1)))
    if warmup:
        ctx.icp_run(warmup)
    ctx.search_stats(True)
    ctx.icp_run(steps)
    phases = ctx.search_phases()
    cand, rows, hard, pts, nodes, leaves, wwaves, longest = ctx.search_stats(False)
out = {"points": n, "steps": steps, "warmup": warmup, "launches": steps,
       "stats": {"candidates": cand, "rows": rows, "lanes_to_hierarchy": hard, "points": pts, "walk_steps": nodes, "walk_leaves": leaves,
                 "walking_waves": wwaves, "longest_walk": longest},
       "phases": phases}
print(json.dumps(out))
