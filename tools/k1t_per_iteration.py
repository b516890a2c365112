#!/usr/bin/env python3
"""Developer tool: the box-hierarchy search time of EACH of the first 52 ICP iterations at the bench size, for the whole moving
cloud and for one rank's eighth of it -- shows how much of the 50-iteration average the early, badly aligned iterations carry."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402


def main():
    capi = load_package().capi
    ctx = capi.Context(0)
    before, after = synth_cloud(np, 1000000)
    for w in (1, 8):
        n = len(before) // w
        ctx.icp_load(before[:n], after, capi.icp_params(eps=0.0, max_iterations=-1))
        ctx.profile_enable(True)
        ctx.profile_select([capi.KERNEL_NN])
        per_it = []
        for _ in range(52):
            ctx.profile_reset()
            ctx.icp_run(1)
            ms, launches = ctx.profile_get(capi.KERNEL_NN)
            per_it.append(round(ms / max(launches, 1), 4))
        ctx.profile_select(None)
        ctx.profile_enable(False)
        print(json.dumps({"ranks": w, "moving_points_per_rank": n, "search_ms_per_iteration": per_it}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
