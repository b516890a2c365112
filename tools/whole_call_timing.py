#!/usr/bin/env python3
"""Developer tool: wall time of whole mi_icp_register calls on host buffers (10 iterations), first and repeated use of a size."""
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
    for n in (500000, 650000, 700000, 775000, 1000000, 1000000, 650000):
        before, after = synth_cloud(np, n)
        p = capi.icp_params(cuda_slam=True, max_iterations=10, eps=0.0, max_distance_squared=10000.0)
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.icp_register(before, after, p)
            times.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter()
        ctx.icp_load(before, after, p)
        ctx.synchronize()
        load_ms = (time.perf_counter() - t0) * 1e3
        print(n, "whole call ms x3:", [round(t, 2) for t in times], "icp_load alone:", round(load_ms, 2), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
