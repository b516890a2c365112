#!/usr/bin/env python3
"""Developer tool: randomized soak of whole registrations against the oracle (oracle/ -- test infrastructure, used here as the checker only):
small random problems (uniform / clustered / planar / duplicated / collinear / coincident clouds), three ICP iterations and five CPD
EM iterations each; prints every case whose result is not finite or farther than 1e-3 (Frobenius, R|t) from the oracle's.
    python tools/reg_soak.py [cases] [seed]"""
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import load_package  # noqa: E402
from nn_soak import cloud  # noqa: E402
import oraclebind as oracle  # noqa: E402


def make(rng, n, kind):
    if kind == 4:                                      # collinear
        d = rng.normal(size=3)
        return (rng.uniform(-5, 5, (n, 1)) * d + rng.normal(size=3)).astype(np.float32)
    if kind == 5:                                      # all points the same
        return np.repeat(rng.uniform(-5, 5, (1, 3)), n, axis=0).astype(np.float32)
    return cloud(rng, n, kind).astype(np.float32)


def problems(cases, seed):
    """reg_soak's problems, in order: (k, degenerate, n, m, kind of the moving cloud, kind of the fixed cloud, moving cloud, fixed cloud).  ONE
    generator consumed case by case -- case k of a seed is only reached through the cases before it (tools/soak_rootcause.py and the pinned
    cases of tests/test_gpu_icp.py replay it the same way)."""
    rng = np.random.default_rng(seed)
    for k in range(cases):
        degenerate = k % 3 == 2              # tiny, collinear or coincident clouds: the rotation is not determined -- only "finite or not" is compared
        if degenerate:
            n = int(10 ** rng.uniform(0.5, 3.2)); m = int(10 ** rng.uniform(0.5, 3.2))
            ks, kt = int(rng.integers(0, 6)), int(rng.integers(0, 6))
        else:
            n = int(10 ** rng.uniform(1.7, 3.3)); m = int(10 ** rng.uniform(1.7, 3.3))
            ks, kt = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        yield k, degenerate, n, m, ks, kt, make(rng, n, ks), make(rng, m, kt)


def frob(R1, t1, R2, t2):
    return float(np.sqrt(((np.asarray(R1, np.float64) - R2) ** 2).sum() + ((np.asarray(t1, np.float64) - t2) ** 2).sum()))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    capi = load_package().capi
    ctx = capi.Context(0)
    worst = {"icp": 0.0, "cpd": 0.0, "hyb": 0.0}
    flagged = 0
    for k, degenerate, n, m, ks, kt, src, tgt in problems(cases, seed):
        Ro, to, ito, eo = oracle.icp(src, tgt, eps=0.0, max_iterations=3)[:4]
        R, t, it, err = ctx.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=3))[:4]
        fin_o = np.isfinite(Ro).all() and np.isfinite(to).all()
        fin = np.isfinite(R).all() and np.isfinite(t).all()
        d = frob(R, t, Ro, to) if fin and fin_o else float("nan")
        scale = max(1.0, float(np.abs(to).max()) if fin_o else 1.0)
        bad = (fin != fin_o) or (not degenerate and ((fin and fin_o and d > 1e-3 * scale) or it != ito))
        if fin and fin_o and not degenerate: worst["icp"] = max(worst["icp"], d / scale)
        if bad:
            flagged += 1
            print("ICP case %d n=%d m=%d kinds %d %d: it %d/%d finite %s/%s diff %.3g" % (k, n, m, ks, kt, it, ito, fin, fin_o, d), flush=True)
        if n >= 2 and m >= 2 and n * m <= 2000000:
            s2 = oracle.cpd_sigma_squared(src, tgt)
            if np.isfinite(s2) and s2 > 0:
                Ro, to, ito, eo = oracle.cpd(src, tgt, eps=0.0, max_iterations=5, tolerance=0.0)[:4]
                sR, t, sc, it, err = ctx.cpd_register(src, tgt, capi.cpd_params(eps=0.0, max_iterations=5, tolerance=0.0, sigma2_init=s2))
                fin_o = np.isfinite(Ro).all() and np.isfinite(to).all()
                fin = np.isfinite(sR).all() and np.isfinite(t).all()
                d = frob(sR, t, Ro, to) if fin and fin_o else float("nan")
                scale = max(1.0, float(np.abs(to).max()) if fin_o else 1.0)
                bad = (fin != fin_o) or (not degenerate and ((fin and fin_o and d > 1e-3 * scale) or it != ito))
                if fin and fin_o and not degenerate: worst["cpd"] = max(worst["cpd"], d / scale)
                if bad:
                    flagged += 1
                    print("CPD case %d n=%d m=%d kinds %d %d: it %d/%d finite %s/%s diff %.3g" % (k, n, m, ks, kt, it, ito, fin, fin_o, d), flush=True)
        # the reference's default approximation (hybrid: FGT E-steps, K-centre sweeps replayed from one E-step to the next), well-posed cases only
        if not degenerate and 60 <= n <= 1500 and 60 <= m <= 1500 and k % 2 == 0:
            s2 = oracle.cpd_sigma_squared(src, tgt)
            if np.isfinite(s2) and s2 > 0:
                Ro, to, ito, eo = oracle.cpd_approx(src, tgt, oracle.APPROX_HYBRID, eps=0.0, max_iterations=6, tolerance=0.0)[:4]
                sR, t, sc, it, err = ctx.cpd_register(src, tgt, capi.cpd_params(eps=0.0, max_iterations=6, tolerance=0.0, sigma2_init=s2,
                                                                               approximation=capi.CPD_APPROX_HYBRID))
                fin_o = np.isfinite(Ro).all() and np.isfinite(to).all()
                fin = np.isfinite(sR).all() and np.isfinite(t).all()
                d = frob(sR, t, Ro, to) if fin and fin_o else float("nan")
                scale = max(1.0, float(np.abs(to).max()) if fin_o else 1.0)
                if fin and fin_o: worst["hyb"] = max(worst["hyb"], d / scale)
                if (fin != fin_o) or (fin and fin_o and d > 2e-3 * scale) or it != ito:
                    flagged += 1
                    print("HYBRID case %d n=%d m=%d kinds %d %d: it %d/%d finite %s/%s diff %.3g" % (k, n, m, ks, kt, it, ito, fin, fin_o, d), flush=True)
    print("registration soak: %d cases, %d flagged, worst relative difference icp %.3g cpd %.3g hybrid %.3g" % (cases, flagged, worst["icp"], worst["cpd"], worst["hyb"]))


if __name__ == "__main__":
    main()
