import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from __graft_entry__ import load_package
from bench import synth_cloud
capi = load_package().capi
before, after = synth_cloud(np, 1000000)
for frac in (1, 8):
    n = len(before) // frac
    with capi.Context(0) as ctx:
        ctx.icp_load(before[:n], after, capi.icp_params(eps=0.0, max_iterations=-1, sync_every=5))
        ctx.profile_enable(True)
        for it in range(0, 30, 5):
            ctx.profile_reset(); ctx.search_stats(True); ctx.icp_run(5)
            cand, rows, hard, pts, nodes, leaves, waves, cyc = ctx.search_stats(False)
            nn = ctx.profile_get(capi.KERNEL_NN)
            print("n %7d it %2d+ nn %.3f ms  walking waves/launch %d  steps/wave %.0f  cycles/wave %.0f  cycles/step %.0f" % (
                n, it, nn[0] / nn[1], waves / 5, (nodes + leaves) / max(waves, 1), cyc / max(waves, 1), cyc / max(nodes + leaves, 1)), flush=True)
