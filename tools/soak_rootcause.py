#!/usr/bin/env python3
"""Developer tool: where do reg_soak's worst "well-posed" differences against the oracle come from?

Regenerates reg_soak's problems (same seeds, same generator) and, for every well-posed case whose 3-iteration ICP or 5-iteration CPD result
lies farther than `floor` (default 1e-4, relative) from the oracle's, runs it again four ways on the same device:
    default                           fp64 sums, K3 in the refined hardware reciprocal / root forms
    MISLAM_SVD_IEEE=1                 K3 in IEEE divisions and roots                    -> closes the gap: K3's fast forms are the cause
    MI_SUM_CPU_SEQUENTIAL (ICP only)  cpu-slam's sequential fp32 centroid / error sums  -> closes the gap: the documented summation deviation
    both
and measures the PROBLEM's own sensitivity with the oracle alone: the oracle on the same clouds with the moving cloud's points REORDERED
(mathematically the same problem; cpu-slam's sequential fp32 sums then round differently) -- if the oracle moves by as much as the device
differs from it, the difference is conditioning of the problem (any rounding anywhere), not an error of a kernel.
    python tools/soak_rootcause.py [cases] [seed] [floor]"""
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from __graft_entry__ import load_package  # noqa: E402
from reg_soak import frob, problems  # noqa: E402
import oraclebind as oracle  # noqa: E402


SENS_RUNS = 12      # reorderings per flagged case (three missed the tail: seed 4 case 202 shows 9e-4 .. 2.9e-2 over twenty)


def contexts(capi):
    os.environ.pop("MISLAM_SVD_IEEE", None)
    fast = capi.Context(0)
    os.environ["MISLAM_SVD_IEEE"] = "1"
    ieee = capi.Context(0)
    os.environ.pop("MISLAM_SVD_IEEE", None)
    return fast, ieee


def cond_of_pairs(src, tgt, R, t):
    """Singular values of the cross-covariance of the oracle's final pairs (fp64): what the Kabsch rotation is read off."""
    cur = src.astype(np.float64) @ np.asarray(R, np.float64).T + np.asarray(t, np.float64)
    idx, _ = oracle.nn_search(cur.astype(np.float32), tgt)
    a = tgt[idx].astype(np.float64)
    b = cur
    H = (a - a.mean(0)).T @ (b - b.mean(0))
    return np.linalg.svd(H, compute_uv=False)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    floor = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-4
    capi = load_package().capi
    fast, ieee = contexts(capi)
    tally = {"icp": [0, 0, 0, 0], "cpd": [0, 0, 0]}       # flagged, closed by IEEE K3, closed by sequential sums, explained by the problem's own sensitivity
    worst_unexplained = {"icp": 0.0, "cpd": 0.0}
    for k, degenerate, n, m, ks, kt, src, tgt in problems(cases, seed):
        if degenerate:
            continue
        prng = np.random.default_rng(1000003 * seed + k)
        # ---- ICP, three iterations
        Ro, to, ito, eo = oracle.icp(src, tgt, eps=0.0, max_iterations=3)[:4]
        if np.isfinite(Ro).all() and np.isfinite(to).all():
            scale = max(1.0, float(np.abs(to).max()))
            runs = {}
            for name, ctx, sm in (("default", fast, capi.SUM_EXACT), ("ieee", ieee, capi.SUM_EXACT), ("seq", fast, capi.SUM_CPU_SEQUENTIAL), ("seq+ieee", ieee, capi.SUM_CPU_SEQUENTIAL)):
                R, t, it, err = ctx.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=3, sum_mode=sm))[:4]
                runs[name] = frob(R, t, Ro, to) / scale if np.isfinite(R).all() else float("nan")
            if runs["default"] > floor:
                sens = 0.0
                for _ in range(SENS_RUNS):      # the same problem, the moving cloud in another order: the oracle against itself
                    Rp, tp = oracle.icp(src[prng.permutation(n)], tgt, eps=0.0, max_iterations=3)[:2]
                    sens = max(sens, frob(Rp, tp, Ro, to) / scale)
                sv = cond_of_pairs(src, tgt, Ro, to)
                tally["icp"][0] += 1
                tally["icp"][1] += runs["ieee"] <= 0.1 * runs["default"]
                tally["icp"][2] += runs["seq"] <= 0.1 * runs["default"]
                explained = sens >= 0.5 * runs["default"] or runs["seq"] <= 0.1 * runs["default"]
                tally["icp"][3] += explained
                if not explained:
                    worst_unexplained["icp"] = max(worst_unexplained["icp"], runs["default"])
                print("ICP case %d n=%d m=%d kinds %d %d: default %.2e | K3 IEEE %.2e | sequential sums %.2e | both %.2e | oracle vs itself reordered %.2e | "
                      "singular values of H %.3g %.3g %.3g%s" % (k, n, m, ks, kt, runs["default"], runs["ieee"], runs["seq"], runs["seq+ieee"], sens, sv[0], sv[1], sv[2],
                                                                "" if explained else "   <-- UNEXPLAINED"), flush=True)
        # ---- CPD, five EM iterations
        if n >= 2 and m >= 2 and n * m <= 2000000:
            s2 = oracle.cpd_sigma_squared(src, tgt)
            if np.isfinite(s2) and s2 > 0:
                Ro, to, ito, eo = oracle.cpd(src, tgt, eps=0.0, max_iterations=5, tolerance=0.0)[:4]
                if np.isfinite(Ro).all() and np.isfinite(to).all():
                    scale = max(1.0, float(np.abs(to).max()))
                    runs = {}
                    for name, ctx in (("default", fast), ("ieee", ieee)):
                        sR, t, sc, it, err = ctx.cpd_register(src, tgt, capi.cpd_params(eps=0.0, max_iterations=5, tolerance=0.0, sigma2_init=s2))
                        runs[name] = frob(sR, t, Ro, to) / scale if np.isfinite(sR).all() else float("nan")
                    if runs["default"] > floor:
                        sens = 0.0
                        for _ in range(SENS_RUNS):      # both clouds reordered: sigma^2_0's, the E-step's and the M-step's sums round differently
                            pb, pa = prng.permutation(n), prng.permutation(m)
                            Rp, tp = oracle.cpd(src[pb], tgt[pa], eps=0.0, max_iterations=5, tolerance=0.0)[:2]
                            sens = max(sens, frob(Rp, tp, Ro, to) / scale)
                        tally["cpd"][0] += 1
                        tally["cpd"][1] += runs["ieee"] <= 0.1 * runs["default"]
                        explained = sens >= 0.5 * runs["default"]
                        tally["cpd"][2] += explained
                        if not explained:
                            worst_unexplained["cpd"] = max(worst_unexplained["cpd"], runs["default"])
                        print("CPD case %d n=%d m=%d kinds %d %d: default %.2e | K3 IEEE %.2e | oracle vs itself reordered %.2e%s"
                              % (k, n, m, ks, kt, runs["default"], runs["ieee"], sens, "" if explained else "   <-- UNEXPLAINED"), flush=True)
    print("soak root cause, seed %d, %d cases, floor %.0e: ICP %d above the floor (%d closed by IEEE K3, %d closed by cpu-slam's sequential sums, %d explained by "
          "the sums or by the problem's own sensitivity; worst unexplained %.2e); CPD %d above the floor (%d closed by IEEE K3, %d explained by the problem's own sensitivity; "
          "worst unexplained %.2e)" % (seed, cases, floor, tally["icp"][0], tally["icp"][1], tally["icp"][2], tally["icp"][3], worst_unexplained["icp"],
                                       tally["cpd"][0], tally["cpd"][1], tally["cpd"][2], worst_unexplained["cpd"]))


if __name__ == "__main__":
    main()
