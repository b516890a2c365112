#!/usr/bin/env python3
"""Developer tool (needs the MISLAM_DEV_WALK_CLOCK variant build: tools/build_variant.sh clock "-DMISLAM_DEV_WALK_CLOCK=1", run with
MISLAM_LIB=.../libmislam_clock.so): how long the walking waves of the search live -- mean time before the walk (prologue + grid
scan), mean walk, and the longest wave from its start to the end of its walk, against the launch's duration.  s_memtime ticks."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402

capi = load_package().capi
before, after = synth_cloud(np, 1000000)
for world in (1, 8):
    ctx = capi.Context(0) if world == 1 else capi.Context(0, 0, world, exchange=lambda arr, kind: None)
    ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=-1, sync_every=1, shard_mode=capi.SHARD_SOURCE))
    ctx.profile_enable(True)
    for it in range(0, 26):
        ctx.profile_reset()
        ctx.search_stats(True)
        ctx.icp_run(1)
        cand, rows, hard, pts, walk, pre, waves, longest = ctx.search_stats(False)
        nn = ctx.profile_get(capi.KERNEL_NN)
        if it % 5 == 0:
            print("ranks %d it %2d  nn %.3f ms  walking waves %5d  before walk %.0f ticks  walk %.0f ticks  longest wave (start -> end of walk) %d ticks" % (
                world, it, nn[0] / nn[1], waves, pre / max(waves, 1), walk / max(waves, 1), longest), flush=True)
    ctx.close()
