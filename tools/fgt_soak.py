#!/usr/bin/env python3
"""Developer tool: randomized soak of the guided K-centre sweep (mi_fgt_kcenter_guided) against the plain sweep: random clouds (uniform,
clustered, planar, duplicated, lattice), sizes up to 2e5, K up to 1500, guesses that are right, truncated, corrupted at a random step,
taken from a transformed copy, or noise -- labels, cell means and choices must be the plain sweep's bit for bit, and `verified` exactly
the length of the common prefix.      python tools/fgt_soak.py [cases] [seed] [--coop]
--coop: the plain sweep comes from a context created under MISLAM_FGT_COOP_SWEEP=0, the guided ones from one under =2 (round 5's cooperative kernel for every sweep
of a cloud beyond 16 384 points)."""
import os
import sys

for _v in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402
from nn_soak import cloud  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = int(argv[0]) if len(argv) > 0 else 60
    rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 20261005)
    capi = load_package().capi
    ctx = ref_ctx = capi.Context(0)
    if "--coop" in sys.argv:         # the plain sweep on a context WITHOUT the cooperative several-workgroup kernel, everything else on one that uses it for every sweep
        os.environ["MISLAM_FGT_COOP_SWEEP"] = "0"
        ref_ctx = capi.Context(0)
        os.environ["MISLAM_FGT_COOP_SWEEP"] = "2"
        ctx = capi.Context(0)
    bad = 0
    for k in range(cases):
        n = int(10 ** rng.uniform(0.5, 5.3))
        K = int(min(n, 10 ** rng.uniform(0, 3.2)))
        kind = int(rng.integers(0, 5))
        if kind == 4:
            g = np.arange(int(round(n ** (1 / 3))) + 1, dtype=np.float32)
            c = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)[:max(n, 2)]
            c = c[rng.permutation(len(c))].astype(np.float32)
        else:
            c = cloud(rng, max(n, 2), kind).astype(np.float32)
        n = len(c)
        K = max(1, min(K, n))
        xc0, lab0 = ref_ctx.fgt_kcenter(c, K)
        xc1, lab1, picked, v = ctx.fgt_kcenter_guided(c, K, np.zeros(0, np.int32))
        ok = v == -1 and np.array_equal(lab0, lab1) and np.array_equal(xc0.view(np.uint32), xc1.view(np.uint32))
        guesses = [(picked, None)]
        if K >= 4:
            cut = int(rng.integers(2, K))
            guesses.append((picked[:cut], None))
            wrong = picked.copy()
            at = int(rng.integers(1, K))
            wrong[at] = int(rng.integers(0, n))
            guesses.append((wrong, None))
            guesses.append((rng.integers(0, n, K).astype(np.int32), None))
            ang = rng.uniform(0, 0.5)
            R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
            moved = (np.float32(rng.uniform(0.8, 1.2)) * (c @ R.T) + rng.normal(size=3).astype(np.float32)).astype(np.float32)
            guesses.append((picked, moved))
        for guess, other in guesses:
            pts = c if other is None else other
            if other is None:
                want_xc, want_lab, want_picked = xc0, lab0, picked
            else:
                want_xc, want_lab, want_picked, _ = ctx.fgt_kcenter_guided(pts, K, np.zeros(0, np.int32))
            xc, lab, pk, ver = ctx.fgt_kcenter_guided(pts, K, guess)
            common = next((i for i in range(min(len(guess), K)) if guess[i] != want_picked[i]), min(len(guess), K))
            good = (np.array_equal(lab, want_lab) and np.array_equal(xc.view(np.uint32), want_xc.view(np.uint32)) and np.array_equal(pk, want_picked)
                    and (ver == common if len(guess) >= 2 else ver == -1))
            ok = ok and good
        bad += 0 if ok else 1
        if not ok or k % 10 == 0:
            print("case %d n=%d K=%d kind=%d: %s" % (k, n, K, kind, "ok" if ok else "MISMATCH"), flush=True)
    print("fgt soak: %d cases, %d mismatches" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
