#!/usr/bin/env python3
"""Developer tool: mi_icp_load + 50 iterations over rising cloud sizes on ONE context -- where a load spends its host time when a size
is new to the context (mi_icp_load_times)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np
from __graft_entry__ import load_package
from bench import synth_cloud
capi = load_package().capi
ctx = capi.Context(0)
ctx.icp_register(*synth_cloud(np, 4096), capi.icp_params(cuda_slam=True, max_iterations=2))
SETS = {"performance": [25000, 50000, 100000, 150000, 275000, 400000, 500000, 525000, 650000, 775000, 900000, 1000000, 1025000, 1150000, 1275000, 1300000],
        "sizes": [1000, 13000, 25000, 37000, 49000, 61000, 73000, 85000, 97000, 10000]}       # testset.cpp:48-117
for k, n in enumerate(SETS[sys.argv[1] if len(sys.argv) > 1 else "performance"]):
    before, after = synth_cloud(np, n, seed=666 + k)
    p = capi.icp_params(cuda_slam=True, max_iterations=50, eps=1e-3, max_distance_squared=10000.0)
    t0 = time.perf_counter(); ctx.icp_load(before, after, p); t1 = time.perf_counter(); ctx.icp_run(-1); ctx.icp_result(); t2 = time.perf_counter()
    lt = ctx.icp_load_times()
    print(n, "load %.2f ms (alloc %.2f, stages %s) run %.2f ms" % ((t1 - t0) * 1e3, lt["workspace"], {k2: round(v, 2) for k2, v in lt.items() if k2 not in ("workspace",)}, (t2 - t1) * 1e3), flush=True)
