#!/usr/bin/env python3
"""Developer tool: rigid CPD at sizes where only the Fast-Gauss-Transform E-step is practical (K9): synthetic uniform clouds of
10^5 and 10^6 points, approximation hybrid, 6 EM iterations, whole mi_cpd_register calls.  One JSON line per case."""
import json
import os
import sys
import time

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from bench import synth_cloud  # noqa: E402


def main():
    capi = load_package().capi
    ctx = capi.Context(0)
    for n in (100000, 1000000):
        before, after = synth_cloud(np, n)
        for label, approx, iters in (("hybrid", capi.CPD_APPROX_HYBRID, 6), ("exact", capi.CPD_APPROX_NONE, 2 if n > 200000 else 6)):
            p = capi.cpd_params(max_iterations=iters, tolerance=0.0, weight=0.1, approximation=approx)
            if label == "hybrid":
                ctx.cpd_register(before, after, p)      # warm-up
            ctx.profile_enable(True)
            ctx.profile_reset()
            t0 = time.perf_counter()
            sR, t, sc, it, err = ctx.cpd_register(before, after, p)
            wall = time.perf_counter() - t0
            prof = {capi.KERNEL_NAMES[k]: ctx.profile_get(k) for k in range(len(capi.KERNEL_NAMES))}
            ctx.profile_enable(False)
            print(json.dumps({"n": n, "approximation": label, "iterations": it, "ms_total": wall * 1e3, "ms_per_em_iteration": wall * 1e3 / max(it, 1),
                              "kernels_ms_per_launch": {k: v[0] / v[1] for k, v in prof.items() if v[1] > 0}, "sigma2": err}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
