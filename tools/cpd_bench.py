#!/usr/bin/env python3
"""Developer tool: times the rigid-CPD path (cfg 4: bunny clouds, exact Gaussian P) with the VALU and the MFMA form of the
P~.[X|1] contraction, plus a larger synthetic case.  Prints one JSON line per case.  Used for DESIGN.md's CPD numbers and as
the command under rocprofv3 for the MFMA counters (profiles/)."""
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
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_cpd.json")))
    cases = [("bunny_14904", z["before"], z["after"], gold["sigma2_init"])]
    if "--big" in sys.argv:
        rng = np.random.default_rng(3)
        b = rng.uniform(-5, 5, (60000, 3)).astype(np.float32)
        a = (b[rng.permutation(60000)] + 0.3).astype(np.float32)
        cases.append(("synthetic_60000", b, a, 0.0))
    ctx = None
    for name, before, after, s2 in cases:
        for mfma in ("1", "0"):
            os.environ["MISLAM_CPD_MFMA"] = mfma               # (read once, at context creation)
            if ctx is not None:
                ctx.close()
            ctx = capi.Context(0)
            p = capi.cpd_params(max_iterations=50, const_scale=0, sigma2_init=s2)
            ctx.cpd_register(before, after, p)          # warm-up (allocations, code load)
            plain = []
            for _ in range(3):                          # the call as a caller makes it: no events on the stream
                t0 = time.perf_counter()
                ctx.cpd_register(before, after, p)
                plain.append((time.perf_counter() - t0) * 1e3)
            ctx.profile_enable(True)
            ctx.profile_reset()
            t0 = time.perf_counter()
            sR, t, scale, it, err = ctx.cpd_register(before, after, p)
            wall = time.perf_counter() - t0
            prof = {capi.KERNEL_NAMES[k]: ctx.profile_get(k) for k in (capi.KERNEL_CPD_DENOM, capi.KERNEL_CPD_CONTRACT, capi.KERNEL_CPD_MSTEP)}
            ctx.profile_enable(False)
            m, n = len(before), len(after)
            pairs = float(m) * n
            den_ms = prof["cpd_denom"][0] / max(prof["cpd_denom"][1], 1)
            con_ms = prof["cpd_contract"][0] / max(prof["cpd_contract"][1], 1)
            print(json.dumps({"case": name, "contraction": "mfma_4x4x1" if mfma == "1" else "valu", "iterations": it,
                              "wall_ms_total": min(plain), "ms_per_em_iteration": min(plain) / max(it, 1), "wall_ms_with_events": wall * 1e3,
                              "K7a_denominator_ms": den_ms, "K7b_contraction_ms": con_ms,
                              "K8_mstep_ms": prof["cpd_mstep"][0] / max(prof["cpd_mstep"][1], 1),
                              "K7a_pairs_per_s": pairs / (den_ms * 1e-3), "K7b_pairs_per_s": pairs / (con_ms * 1e-3),
                              "t": [float(x) for x in t], "sigma2": err}), flush=True)
        # the reference's approximate modes (K9: FGT E-step on the device), same clouds
        os.environ.pop("MISLAM_CPD_MFMA", None)
        ctx.close()
        ctx = capi.Context(0)
        for approx, label in ((capi.CPD_APPROX_HYBRID, "hybrid"), (capi.CPD_APPROX_FULL, "full")):
            p = capi.cpd_params(max_iterations=50 if label == "hybrid" else 17, const_scale=0, sigma2_init=s2, approximation=approx)
            ctx.cpd_register(before, after, p)
            plain = []
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.cpd_register(before, after, p)
                plain.append((time.perf_counter() - t0) * 1e3)
            ctx.profile_enable(True)
            ctx.profile_reset()
            t0 = time.perf_counter()
            sR, t, scale, it, err = ctx.cpd_register(before, after, p)
            wall = time.perf_counter() - t0
            fgt_ms, fgt_n = ctx.profile_get(capi.KERNEL_CPD_FGT)
            ctx.profile_enable(False)
            print(json.dumps({"case": name, "approximation": label, "iterations": it, "wall_ms_total": min(plain), "wall_ms_with_events": wall * 1e3,
                              "ms_per_em_iteration": min(plain) / max(it, 1), "fgt_esteps": fgt_n,
                              "K9_fgt_estep_ms": fgt_ms / max(fgt_n, 1), "t": [float(x) for x in t], "sigma2": err}), flush=True)
    ctx.close()
    # Round 5: the hybrid mode's truncated E-step alone (K7t, cpd_trunc.hip: culled by tile boxes) against round 4's every-pair truncated kernels
    # (MISLAM_CPD_TRUNC_CULL=0), on the bunny clouds at the sigma^2 where the hybrid mode switches over and late in a run; and a hybrid
    # registration at 1e5 points (--hybrid-1e5): the size where culling decides whether the mode is usable at all
    for cull in ("1", "0"):
        os.environ["MISLAM_CPD_TRUNC_CULL"] = cull
        ctx = capi.Context(0)
        name, before, after, s2 = cases[0]
        for sigma2 in (0.05, 0.01, 0.001):
            c = 1.0
            ctx.cpd_estep_truncated(before, after, c, sigma2, 1e-3)
            ctx.profile_enable(True)
            ctx.profile_reset()
            for _ in range(5):
                ctx.cpd_estep_truncated(before, after, c, sigma2, 1e-3)
            den, con = ctx.profile_get(capi.KERNEL_CPD_DENOM), ctx.profile_get(capi.KERNEL_CPD_CONTRACT)
            ctx.profile_enable(False)
            print(json.dumps({"case": name, "truncated_estep": "culled (K7t)" if cull == "1" else "every pair (round 4)", "sigma2": sigma2,
                              "K7a_ms": den[0] / max(den[1], 1), "K7b_ms": con[0] / max(con[1], 1),
                              "K7a_plus_K7b_ms": den[0] / max(den[1], 1) + con[0] / max(con[1], 1)}), flush=True)
        if "--hybrid-1e5" in sys.argv:
            # a SURFACE (a bumpy sphere, like the reference's scanned models), the headline's rigid motion, a little noise: the registration
            # converges, sigma^2 falls through the hybrid mode's switch and the truncated E-step runs for most of the iterations
            rng = np.random.default_rng(5)
            u = rng.normal(size=(100000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
            b5 = (5.0 * u * (1.0 + 0.3 * np.sin(3 * u[:, :1]) * np.cos(2 * u[:, 1:2]))).astype(np.float32)
            axis = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
            K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
            Rm = np.eye(3) + np.sin(0.2) * K + (1 - np.cos(0.2)) * (K @ K)
            a5 = (b5[rng.permutation(100000)].astype(np.float64) @ Rm.T + 1.0 + rng.normal(scale=0.01, size=(100000, 3))).astype(np.float32)
            p = capi.cpd_params(max_iterations=40, eps=0.0, tolerance=0.0, approximation=capi.CPD_APPROX_HYBRID)
            ctx.cpd_register(b5, a5, p)
            t0 = time.perf_counter()
            sR, t, scale, it, err = ctx.cpd_register(b5, a5, p)
            wall = (time.perf_counter() - t0) * 1e3
            print(json.dumps({"case": "bumpy_sphere_100000", "approximation": "hybrid", "truncated_estep": "culled (K7t)" if cull == "1" else "every pair (round 4)",
                              "iterations": it, "wall_ms_total": wall, "ms_per_em_iteration": wall / max(it, 1), "sigma2": err, "t": [float(x) for x in t]}), flush=True)
        ctx.close()
    os.environ.pop("MISLAM_CPD_TRUNC_CULL", None)


if __name__ == "__main__":
    main()
