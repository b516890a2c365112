#!/usr/bin/env python3
"""Developer tool: randomized soak of round 5's FGT paths through whole registrations.  Random surface / volume clouds of 3 000 ... 300 000 points, a random
rigid motion, hybrid and full rigid CPD for a few EM iterations on two contexts: the default one (member lists inside the model kernel up to 32 768 points,
cooperative K-centre sweep beyond 16 384, big cells split over workgroups, the fixed cloud's clustering on a second stream) and one with every one of these
switched off (MISLAM_FGT_LISTS_IN_MODEL=0 MISLAM_FGT_COOP_SWEEP=0 MISLAM_FGT_MODEL_SPLITS=0 MISLAM_FGT_TWO_STREAMS=0: rounds 1-4).  Lists, sweep and streams
change no bit; the split model build changes the summation order of a big cell's coefficients: the same iteration count, s*R|t within 2e-5.
    python tools/fgt_paths_soak.py [cases] [seed]"""
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

OFF = {"MISLAM_FGT_LISTS_IN_MODEL": "0", "MISLAM_FGT_COOP_SWEEP": "0", "MISLAM_FGT_MODEL_SPLITS": "0", "MISLAM_FGT_TWO_STREAMS": "0"}


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    capi = load_package().capi
    new = capi.Context(0)
    os.environ.update(OFF)
    old = capi.Context(0)
    for k in OFF:
        del os.environ[k]
    bad, worst = 0, 0.0
    for k in range(cases):
        n = int(10 ** rng.uniform(3.5, 5.5))
        m = max(1000, int(n * rng.uniform(0.5, 1.2)))
        if rng.random() < 0.5:      # a surface
            u = rng.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
            b = (5.0 * u * (1.0 + 0.3 * np.sin(3 * u[:, :1]) * np.cos(2 * u[:, 1:2]))).astype(np.float32)
        else:                       # a volume
            b = rng.uniform(-5, 5, (m, 3)).astype(np.float32)
        axis = rng.normal(size=3); axis /= np.linalg.norm(axis)
        ang = rng.uniform(0.05, 0.4)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        a = (b[rng.integers(0, m, n)].astype(np.float64) @ R.T + rng.uniform(-1, 1, 3) + rng.normal(scale=0.01, size=(n, 3))).astype(np.float32)
        for approx, label, iters in ((capi.CPD_APPROX_HYBRID, "hybrid", 5), (capi.CPD_APPROX_FULL, "full", 4)):
            p = capi.cpd_params(max_iterations=iters, eps=0.0, tolerance=0.0, approximation=approx, weight=float(rng.choice([0.1, 0.3])))
            x, y = new.cpd_register(b, a, p), old.cpd_register(b, a, p)
            x2 = new.cpd_register(b, a, p)          # and again on the same context: buffers, clusterings and guesses carried over
            d = float(np.sqrt(((x[0] - y[0]) ** 2).sum() + ((x[1] - y[1]) ** 2).sum()))
            fin = bool(np.isfinite(x[0]).all() and np.isfinite(x[1]).all())
            same_again = x2[3] == x[3] and np.array_equal(x2[0], x[0]) and np.array_equal(x2[1], x[1])
            ok = fin and x[3] == y[3] and d < 2e-5 * max(1.0, float(np.abs(y[1]).max())) and same_again
            worst = max(worst, d)
            if not ok:
                bad += 1
            if not ok or k % 5 == 0:
                print("case %d m=%d n=%d %s: iterations %d/%d, |d(sR|t)|_F %.2e, repeatable %s%s" % (k, m, n, label, x[3], y[3], d, same_again, "" if ok else "   <-- MISMATCH"), flush=True)
    print("fgt paths soak: %d cases x 2 modes, %d mismatches, worst |d| %.2e" % (cases, bad, worst))


if __name__ == "__main__":
    main()
