"""TEST INFRASTRUCTURE ONLY.  Fixtures for the ICP leg of the reference's convergence test set (GetConvergenceTestSet, source/common/testset.cpp:119-187:
cloud-spread 10, max-iterations 100, max-distance-squared 10000, parallel policy, nine (rotation range, translation range) pairs -- 0.2 / 0.4 / 0.6 rad
x 10 / 20 / 30 units -- with a RANDOM known transformation drawn by the reference's own generators) at the set's first size, 20 000 points of
bird.obj (testset.cpp:19-38: "<= 35 008 -> bird").  The reference's own code (oracle/_ref) prepares the clouds (GetCloudsFromConfig incl. the random
transformation: ref_clouds_from_config_random) and runs cpu-slam's ICP (basicicp.cpp:23-61); the set has no seeds (std::random_device): the
fixture fixes one per configuration.  Run in the build container (~2 min):

    python oracle/make_golden_convergence.py

Output  tests/golden/convergence_icp.json   per configuration: the JSON configuration file, sizes and sha256 of the prepared clouds, the transformation
        the reference drew, cpu-slam's iterations / R / t / error, the restatement's (oracle/slam_oracle.c) beside it.
The raw cloud comes from tests/golden/noise_meshes.npz (bird.obj's vertex table and face corners).
"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import oraclebind as O      # noqa: E402
from oracle import refbind as ref       # noqa: E402

SIZE = 20000
PAIRS = [(0.2, 10.0), (0.4, 10.0), (0.6, 10.0), (0.2, 20.0), (0.4, 20.0), (0.6, 20.0), (0.2, 30.0), (0.4, 30.0), (0.6, 30.0)]   # testset.cpp:146-176


def frob(R1, t1, R2, t2):
    return float(np.sqrt(((np.asarray(R1, np.float64) - R2) ** 2).sum() + ((np.asarray(t1, np.float64) - t2) ** 2).sum()))


def main():
    z = np.load(os.path.join(GOLD, "noise_meshes.npz"))
    raw = np.ascontiguousarray(z["bird_v"][z["bird_f"].astype(np.int64)])
    assert len(raw) == 35008
    out = []
    devnull, keep = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    for k, (rot, trans) in enumerate(PAIRS):
        seed = 1000 + k
        cfg = {"before-path": "data/bird.obj", "after-path": "data/bird.obj", "method": "icp", "policy": "parallel", "max-iterations": 100,
               "cloud-spread": 10.0, "max-distance-squared": 10000.0, "rotation-range": rot, "translation-range": trans,
               "cloud-before-resize": SIZE, "cloud-after-resize": SIZE, "random-seed": seed}
        before, after, R_known, t_known = ref.clouds_from_config_random(raw, None, seed, rot, trans, resize_before=SIZE, resize_after=SIZE, spread=10.0)
        os.dup2(devnull, 1)
        try:
            t0 = time.time()
            R, t, it, err = ref.icp(before, after, eps=1e-3, max_distance_squared=10000.0, max_iterations=100, parallel=True)
            dt = time.time() - t0
            Ro, to, ito, eo = O.icp(before, after, 1e-3, 10000.0, 100)[:4]
            # cpu-slam against itself: the same two point sets in another order (its sequential fp32 centroid and error sums round differently)
            rng = np.random.default_rng(seed)
            Rp, tp, itp, errp = ref.icp(before[rng.permutation(len(before))], after[rng.permutation(len(after))], eps=1e-3, max_distance_squared=10000.0,
                                        max_iterations=100, parallel=True)
        finally:
            os.dup2(keep, 1)
        out.append({"config_json": cfg, "rotation_range": rot, "translation_range": trans, "seed": seed, "n_before": len(before), "n_after": len(after),
                    "sha256_before": hashlib.sha256(before.tobytes()).hexdigest(), "sha256_after": hashlib.sha256(after.tobytes()).hexdigest(),
                    "known_R": np.asarray(R_known, np.float64).tolist(), "known_t": np.asarray(t_known, np.float64).tolist(),
                    "cpu_slam": {"iterations": int(it), "R": np.asarray(R, np.float64).tolist(), "t": np.asarray(t, np.float64).tolist(), "error": float(err), "seconds": round(dt, 1)},
                    "oracle": {"iterations": int(ito), "R": np.asarray(Ro, np.float64).tolist(), "t": np.asarray(to, np.float64).tolist(), "error": float(eo)},
                    "oracle_vs_cpu_slam": frob(Ro, to, R, t), "cpu_slam_vs_known": frob(R, t, R_known, t_known),
                    "cpu_slam_reordered": {"iterations": int(itp), "distance": frob(Rp, tp, R, t), "error": float(errp)}})
        print("rot %.1f trans %2.0f: cpu-slam %3d iterations, error %.4g, |d| to the known motion %.3g; restatement %3d iterations, |d| vs cpu-slam %.2e; "
              "cpu-slam reordered %3d iterations, |d| vs cpu-slam %.2e (%.0f s)" % (rot, trans, it, err, out[-1]["cpu_slam_vs_known"], ito, out[-1]["oracle_vs_cpu_slam"],
                                                                                 itp, out[-1]["cpu_slam_reordered"]["distance"], dt), flush=True)
    with open(os.path.join(GOLD, "convergence_icp.json"), "w") as f:
        json.dump({"source": "GetConvergenceTestSet (testset.cpp:119-187), ICP, 20 000 points of bird.obj, run by oracle/_ref; see oracle/make_golden_convergence.py", "configs": out}, f, indent=1)
    print("wrote tests/golden/convergence_icp.json")


if __name__ == "__main__":
    main()
