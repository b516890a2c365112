#!/usr/bin/env python3
"""Developer tool: times the non-iterative registration ("method": "nicp") on the bunny clouds and on a 10^6-point synthetic pair,
whole mi_nicp_register calls on host buffers (upload, moments pass, all repetitions, subcloud scoring).  One JSON line per case."""
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


def main():
    capi = load_package().capi
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny_clouds.npz"))
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_nicp.json")))
    rng = np.random.default_rng(3)
    big_b = rng.uniform(-5, 5, (1000000, 3)).astype(np.float32) * np.array([1.0, 0.6, 0.3], np.float32)
    ang = 0.2
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    big_a = (big_b @ Rz.T + np.array([1.0, 2.0, 3.0], np.float32)).astype(np.float32)
    cases = [("bunny_14904", z["before"], z["after"]), ("synthetic_1000000", big_b, big_a)]
    ctx = capi.Context(0)
    for name, before, after in cases:
        n = len(before)
        reps, sub_n = 32, 1000                                   # the parser's defaults (configparser.cpp:234-236)
        prng = np.random.default_rng(1)
        sub = prng.permutation(n)[:sub_n].astype(np.int32)
        heads = np.stack([prng.permutation(n)[:3] for _ in range(reps)]).astype(np.int32)
        for label, approx in (("none", 0), ("hybrid", 2)):
            p = capi.nicp_params(eps=0.0, max_repetitions=reps, approximation=approx)      # eps = 0: no early exit, all 32 repetitions
            ctx.nicp_register(before, after, p, heads, sub)      # warm-up
            ctx.profile_enable(True)
            ctx.profile_reset()
            t0 = time.perf_counter()
            R, t, it, err = ctx.nicp_register(before, after, p, heads, sub)
            wall = time.perf_counter() - t0
            prof = {capi.KERNEL_NAMES[k]: ctx.profile_get(k) for k in range(len(capi.KERNEL_NAMES))}
            ctx.profile_enable(False)
            print(json.dumps({"case": name, "approximation": label, "repetitions": it, "ms_total": wall * 1e3,
                              "kernels_ms_per_launch": {k: v[0] / v[1] for k, v in prof.items() if v[1] > 0},
                              "nn_launches": prof["nn"][1], "error": err}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
