#!/usr/bin/env python3
"""Developer tool: randomized soak of the box-hierarchy and cell-grid searches against the every-pair search (bit-exact indices and distances):
random sizes up to 3e5, uniform / clustered / planar / duplicated clouds, both distance arithmetics, repeated searches with
moving sources so that warm starts (the seeds a previous search leaves) are exercised through ICP as well."""
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):   # see bench.quiet_host_pools: BLAS pools vs the CPU quota
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def cloud(rng, n, kind):
    if kind == 0:
        return rng.uniform(-5, 5, (n, 3))
    if kind == 1:
        c = rng.uniform(-5, 5, (rng.integers(2, 9), 3))
        return c[rng.integers(0, len(c), n)] + rng.normal(scale=rng.uniform(0.01, 0.5), size=(n, 3))
    if kind == 2:
        p = rng.uniform(-5, 5, (n, 3))
        p[:, rng.integers(0, 3)] = rng.uniform(-1e-3, 1e-3, n) if rng.random() < 0.5 else 0.25
        return p
    base = rng.uniform(-5, 5, (max(n // 3, 1), 3))
    return base[rng.integers(0, len(base), n)]          # heavy duplication: ties everywhere


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    capi = load_package().capi
    ctx = capi.Context(0)
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 20261004)
    bad = 0
    for k in range(cases):
        n = int(10 ** rng.uniform(0, 5.48))
        m = int(10 ** rng.uniform(0.5, 5.48))
        tgt = cloud(rng, m, rng.integers(0, 4)).astype(np.float32)
        src = cloud(rng, n, rng.integers(0, 4)).astype(np.float32)
        if rng.random() < 0.3:
            take = rng.integers(0, m, min(n, m))
            src[:len(take)] = tgt[take]                  # exact hits
        mode = int(rng.integers(0, 2))
        a = ctx.nn_search(src, tgt, mode, capi.NN_BRUTEFORCE)
        ok = True
        for indexed in (capi.NN_TREE, capi.NN_GRID):
            b = ctx.nn_search(src, tgt, mode, indexed)
            ok = ok and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        if ok and k % 4 == 0 and n >= 10 and m >= 10:    # a short ICP through every search (the grid's is the fused iteration):
            runs = []                                    # bitwise the same trajectory
            for nn in (capi.NN_BRUTEFORCE, capi.NN_TREE, capi.NN_GRID):
                runs.append(ctx.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=4, nn_mode=nn, dist_mode=mode)))
            r1 = runs[0]
            ok = all(r1[2] == r2[2] and r1[3] == r2[3] and np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1]) for r2 in runs[1:])
        bad += 0 if ok else 1
        if not ok or k % 20 == 0:
            print("case %d n=%d m=%d mode=%d: %s" % (k, n, m, mode, "ok" if ok else "MISMATCH"), flush=True)
    print("soak: %d cases, %d mismatches" % (cases, bad))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
