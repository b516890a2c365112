#!/usr/bin/env python3
"""Developer tool: is the occasional 20-70 ms stall of a whole mi_icp_register call an idle-GPU effect?  Same clouds, same sizes
(no buffer growth after the first call), with and without a host-side pause before the call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402


def main():
    capi = load_package().capi
    ctx = capi.Context(0)
    before, after = synth_cloud(np, 1000000)
    p = capi.icp_params(cuda_slam=True, max_iterations=10, eps=0.0, max_distance_squared=10000.0)
    ctx.icp_register(before, after, p)
    for pause in (0.0, 0.0, 0.2, 0.2, 1.0, 1.0, 3.0, 3.0, 0.0, 0.0):
        time.sleep(pause)
        t0 = time.perf_counter()
        ctx.icp_register(before, after, p)
        print("pause %.1f s -> whole call %.2f ms" % (pause, (time.perf_counter() - t0) * 1e3), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
