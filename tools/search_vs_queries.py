#!/usr/bin/env python3
"""Developer tool: what one rank of a W-GPU run sees -- the default (cell-grid) search for its share of the moving cloud against the
whole fixed cloud (10^6 and 10^7 points; the moving cloud is what the ranks split), 20 iterations after 5
warm-up ones.  Rank 0 of a W-rank context whose "all-reduce" is a no-op: the library deals it the 64-point chunks 0, W, 2W ... of
the Hilbert-ordered cloud exactly as in a real run; its moments are then those of its own share only (the registration still
converges: the share is a uniform sample), which is all the timing needs.
    python tools/search_vs_queries.py [points ...]"""
import json
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402


def main():
    capi = load_package().capi
    for points in [int(a) for a in sys.argv[1:]] or [1000000, 10000000]:
        before, after = synth_cloud(np, points)
        for w in (1, 2, 4, 8):
            ctx = capi.Context(0) if w == 1 else capi.Context(0, 0, w, exchange=lambda arr, kind: None)
            ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=-1, shard_mode=capi.SHARD_SOURCE))
            ctx.icp_run(5)
            ctx.profile_enable(True)
            ctx.profile_select([capi.KERNEL_NN])
            ctx.profile_reset()
            ctx.icp_run(20)
            ms, launches = ctx.profile_get(capi.KERNEL_NN)
            ctx.profile_select(None)
            ctx.profile_reset()
            ctx.icp_run(5)
            rest = sum(ctx.profile_get(k)[0] / 5 for k in (capi.KERNEL_MOMENTS, capi.KERNEL_SOLVE, capi.KERNEL_TRANSFORM, capi.KERNEL_FINALIZE))
            ctx.profile_enable(False)
            print(json.dumps({"points": points, "ranks": w, "moving_points_per_rank": points // w, "search_ms": ms / launches, "other_kernels_ms": rest}), flush=True)
            ctx.close()


if __name__ == "__main__":
    main()
