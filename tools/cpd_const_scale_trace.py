#!/usr/bin/env python3
"""Developer tool: "cpd-const-scale": true on the bunny clouds, EM iteration by EM iteration -- the device (the run capped at
k = 1, 2, ... iterations: every capped run retraces the same trajectory) beside the oracle's trace: sigma^2 and the distance of s*R|t.  Shows where
the two leave each other and by how much per iteration; `MISLAM_SVD_IEEE=1` tells whether K3's fast forms have a part in it.
    python tools/cpd_const_scale_trace.py [max_iterations]"""
import json
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import load_package  # noqa: E402
import oraclebind as oracle  # noqa: E402


def frob(R1, t1, R2, t2):
    return float(np.sqrt(((np.asarray(R1, np.float64) - R2) ** 2).sum() + ((np.asarray(t1, np.float64) - t2) ** 2).sum()))


def main():
    cap = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    capi = load_package().capi
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny_clouds.npz"))
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_cpd.json")))
    before, after = z["before"], z["after"]
    ctx = capi.Context(0)
    os.environ["MISLAM_SVD_IEEE"] = "1"
    ctx_ieee = capi.Context(0)
    os.environ.pop("MISLAM_SVD_IEEE")
    for const_scale in (1, 0):
        # the oracle's trace needs its own sigma^2_0 = the fixture's (cpu-slam's saturated sum): oracle.cpd computes exactly that
        out = oracle.cpd(before, after, eps=1e-3, weight=0.3, const_scale=bool(const_scale), max_iterations=cap, tolerance=1e-3, trace_cap=cap)
        Ro, to, ito, eo, trace = out
        print("const_scale %d: oracle %d iterations, final sigma^2 %.6g" % (const_scale, ito, eo))
        for k in range(1, ito + 1):
            row = trace[k - 1]
            Rk = row[4:13].reshape(3, 3).T * row[3]       # column-major R, times the scale
            tk = row[13:16]
            line = "  it %2d oracle sigma^2 %.6e" % (k, row[0])
            for name, c in (("device", ctx), ("device, IEEE K3", ctx_ieee)):
                sR, t, sc, it, err = c.cpd_register(before, after, capi.cpd_params(max_iterations=k, const_scale=const_scale, sigma2_init=g["sigma2_init"]))
                line += " | %s sigma^2 %.6e (rel %.1e) |d| %.2e" % (name, err, abs(err - row[0]) / max(row[0], 1e-30), frob(sR, t, Rk, tk))
            print(line, flush=True)


if __name__ == "__main__":
    main()
