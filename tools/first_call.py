#!/usr/bin/env python3
"""Developer tool: ONE first registration call at 10^6 points (context creation excluded), for a HIP API trace of what the first
call of a size spends outside the kernels."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402

capi = load_package().capi
ctx = capi.Context(0)
small_b, small_a = synth_cloud(np, 1000)
ctx.icp_register(small_b, small_a, capi.icp_params(max_iterations=2))      # runtime warm-up at a tiny size
before, after = synth_cloud(np, 1000000)
p = capi.icp_params(cuda_slam=True, max_iterations=10, eps=0.0, max_distance_squared=10000.0)
for k in range(2):
    t0 = time.perf_counter()
    ctx.icp_register(before, after, p)
    print("call %d: %.2f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True)
ctx.close()
