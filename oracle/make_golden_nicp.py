"""TEST INFRASTRUCTURE ONLY.  Generates the non-iterative-registration fixtures under tests/golden/ by running the reference's OWN
CPU code (oracle/_ref: source/cpu-slam/noniterative.cpp with the sequential policy, seeded like clouds-from-config seeds
Common::mtRandom) on the committed bunny clouds.  Run in the build container:

    python oracle/make_golden_nicp.py

Outputs
    tests/golden/bunny_nicp.json   per approximation type (none / full / hybrid): the reference's R, t, repetitions, error; the
                                   first three entries of every repetition's permutation and the subcloud indices -- what the
                                   product ABI takes -- drawn from the reference's generator
    tests/golden/bunny_nicp_perms.npz   the full permutations (uint16) of the first repetitions, for the restatement, which sums
                                   in permuted order like the reference
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import refbind as ref          # noqa: E402

SEED, REPS, SUB = 666, 8, 1000


def main():
    z = np.load(os.path.join(GOLD, "bunny_clouds.npz"))
    before, after = z["before"], z["after"]
    n = len(before)
    sub_perm = ref.random_permutation(SEED, n, 0)                       # GetSubcloud draws first (common.cpp:30)
    perms = np.stack([ref.random_permutation(SEED, n, 1 + k) for k in range(REPS)])
    out = dict(seed=SEED, repetitions=REPS, subcloud_size=SUB, eps=1e-3, subcloud_idx=sub_perm[:SUB].tolist(),
               order_heads=perms[:, :3].tolist(), runs={})
    for name, approx in (("none", 0), ("full", 1), ("hybrid", 2)):
        R, t, reps, err = ref.nicp(before, after, 1e-3, REPS, approx, False, SUB, SEED)
        out["runs"][name] = dict(R=R.tolist(), t=t.tolist(), repetitions=reps, error=err)
    singles = []
    for k in range(3):                                                  # GetSingleNonIterativeSlamResult on permuted clouds
        R, t, e = ref.nicp_single(before[perms[k]], after[perms[k]])
        singles.append(dict(R=R.tolist(), t=t.tolist(), approximated_error=e))
    out["singles"] = singles
    json.dump(out, open(os.path.join(GOLD, "bunny_nicp.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(GOLD, "bunny_nicp_perms.npz"), perms=perms.astype(np.uint16), sub_perm=sub_perm[:SUB].astype(np.uint16))
    print("wrote bunny_nicp.json, bunny_nicp_perms.npz")


if __name__ == "__main__":
    main()
