#!/usr/bin/env python3
"""Developer tool: the fused search on two AXIS-ALIGNED cubes that overlap only partly -- the moving cloud hangs out of the fixed cloud's bounding box
by a few cells on three faces (the case nn_grid.hip's grid_lane_cap2 serves from the grid since round 5; bench.py's rotated cube keeps its
"outside" points INSIDE the axis-aligned box of the fixed cloud, in the empty wedges).  Search time and the share of lanes that walk the hierarchy,
per group of iterations.  A/B: MISLAM_LIB=cuda-slam_amd/variants/libmislam_NAME.so (tools/build_variant.sh NAME -DMISLAM_GRID_EXTENT_REACH=0).
    python tools/outside_probe.py [points] [shift in cells]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

capi = load_package().capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
shift_cells = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
rng = np.random.default_rng(9)
fixed = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
h = 10.0 / (n / 1.25) ** (1.0 / 3.0)
# the moving cloud: the same cube pushed out of the fixed one by `shift_cells` cells on x, 0.6 of that on y, 0.3 on z, its own points
moving = (rng.uniform(-5, 5, (n, 3)) + shift_cells * h * np.array([1.0, 0.6, 0.3])).astype(np.float32)
with capi.Context(0) as ctx:
    # eps = 0 and ONE iteration per group re-registered from scratch would converge; instead: searches of the SAME configuration, the registration held
    # still by max_iterations = 1 reloads -- what is timed is the search at this overlap
    for rep in range(3):
        ctx.icp_load(moving, fixed, capi.icp_params(eps=0.0, max_iterations=-1, sync_every=2))
        ctx.search_stats(True)           # (the counting build of the kernel: for the shares only)
        ctx.icp_run(2)                   # iteration 0 (no starting candidate) and iteration 1 (previous matches, the cloud barely moved)
        cand, rows, hard, pts, nodes, leaves, waves, cyc = ctx.search_stats(False)
        ctx.icp_load(moving, fixed, capi.icp_params(eps=0.0, max_iterations=-1, sync_every=2))
        ctx.profile_enable(True)
        ctx.profile_select([capi.KERNEL_NN])
        ctx.profile_reset()
        ctx.icp_run(2)
        nn = ctx.profile_get(capi.KERNEL_NN)
        ctx.profile_enable(False)
        print("shift %.1f cells, %d points: search %.3f ms per iteration (2 iterations), lanes walking %.2f %%, candidates per point %.1f" % (
            shift_cells, n, nn[0] / max(nn[1], 1), 100.0 * hard / max(pts, 1), cand / max(pts, 1)), flush=True)
