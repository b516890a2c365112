#!/usr/bin/env python3
"""Developer tool: the device input stage (mi_prepare_cloud: normalise to a spread, shuffle, 10 % noisy points, 100 outliers, known
transformation) against the CPU restatement of the reference's stage on the same draws, whole call, host buffers in and out."""
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
from oracle import oraclebind as O  # noqa: E402


def main():
    capi = load_package().capi
    ctx = capi.Context(0)
    R = np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32)
    t = np.array([1, 2, 3], np.float32)
    for n in (100000, 1000000, 10000000):
        rng = np.random.default_rng(n)
        raw = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
        k = n // 10
        kw = dict(shuffle_idx=rng.permutation(n).astype(np.int32), noise_rows=np.sort(rng.permutation(n)[:k]).astype(np.int32),
                  noise_unit=rng.uniform(0, 1, (k, 3)).astype(np.float32), noise_intensity=0.05,
                  outlier_unit=rng.uniform(0, 1, (100, 3)).astype(np.float32), spread=10.0, R=R, t=t)
        ctx.prepare_cloud(raw, **kw)
        t0 = time.perf_counter()
        got = ctx.prepare_cloud(raw, **kw)
        dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        want = O.prepare_cloud(raw, **kw)
        cpu = time.perf_counter() - t0
        print(json.dumps({"points": n, "device_ms": dev * 1e3, "cpu_restatement_ms": cpu * 1e3, "bit_exact": bool(np.array_equal(got, want))}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
