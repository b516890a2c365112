"""TEST INFRASTRUCTURE ONLY.  Fixtures for the NICP legs of the reference's test sets (source/common/testset.cpp) at the sizes whose .obj files exist:

  * GetSizesTestSet(NoniterativeIcp)  :48-80   approximation none, sequential policy, no cloud-spread, parser defaults otherwise (32 repetitions, subcloud 1 000,
                                               eps 1e-3): 1 000 / 5 000 / 13 000 points of bunny.obj, 33 000 of bird.obj
  * GetPerformanceTestSet(NoniterativeIcp) :82-116  hybrid, sequential, cloud-spread 10, nicp-subcloud-size 1 000, nicp-iterations 64: 10 000 points of bunny.obj,
                                               20 000 / 30 000 of bird.obj
each with the random known transformation of (0.2 rad, 10 units) the reference's generators draw.  The reference's own code (oracle/_ref) runs what its program
runs: GetCloudsFromConfig (ref_clouds_from_config_random) and then, ON THE GENERATOR AS THE INPUT STAGE LEFT IT, cpu-slam's
GetNonIterativeTransformationMatrix (noniterative.cpp:284-290, sequential policy: ref_nicp_continue).  The sets have no seeds (std::random_device): the fixture
fixes one per configuration.  Run in the build container (~1 min):

    python oracle/make_golden_nicp_sets.py

Output  tests/golden/nicp_sets.json   per configuration: the JSON configuration file, sizes and sha256 of the prepared clouds, the transformation drawn,
        cpu-slam's repetitions / R / t / error.  The raw clouds come from tests/golden/noise_meshes.npz.
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

from oracle import refbind as ref       # noqa: E402

SIZES_SET = [(1000, "bunny"), (5000, "bunny"), (13000, "bunny"), (33000, "bird")]
PERFORMANCE_SET = [(10000, "bunny"), (20000, "bird"), (30000, "bird")]


def frob(R1, t1, R2, t2):
    return float(np.sqrt(((np.asarray(R1, np.float64) - R2) ** 2).sum() + ((np.asarray(t1, np.float64) - t2) ** 2).sum()))


def main():
    z = np.load(os.path.join(GOLD, "noise_meshes.npz"))
    raws = {name: np.ascontiguousarray(z[name + "_v"][z[name + "_f"].astype(np.int64)]) for name in ("bunny", "bird")}
    out = []
    devnull, keep = os.open(os.devnull, os.O_WRONLY), os.dup(1)
    todo = [("sizes", s, n) for s, n in SIZES_SET] + [("performance", s, n) for s, n in PERFORMANCE_SET]
    for j, (which, size, name) in enumerate(todo):
        seed = 3000 + j
        cfg = {"before-path": "data/%s.obj" % name, "after-path": "data/%s.obj" % name, "method": "nicp", "policy": "sequential", "max-iterations": 50,
               "max-distance-squared": 10000.0, "rotation-range": 0.2, "translation-range": 10.0, "cloud-before-resize": size, "cloud-after-resize": size,
               "random-seed": seed, "cpd-weight": 0.1}
        if which == "sizes":
            cfg.update({"approximation-type": "none"})
            spread, approx, reps, sub = None, 0, 32, 1000
        else:
            cfg.update({"approximation-type": "hybrid", "cloud-spread": 10.0, "nicp-subcloud-size": 1000, "nicp-iterations": 64})
            spread, approx, reps, sub = 10.0, 2, 64, 1000
        before, after, R_known, t_known = ref.clouds_from_config_random(raws[name], None, seed, 0.2, 10.0, resize_before=size, resize_after=size, spread=spread)
        os.dup2(devnull, 1)
        try:
            t0 = time.time()
            R, t, nrep, err = ref.nicp_continue(before, after, 1e-3, reps, approx, False, sub)      # (behind the input stage's draws: no reseeding)
            dt = time.time() - t0
        finally:
            os.dup2(keep, 1)
        out.append({"set": which, "config_json": cfg, "seed": seed, "n_before": len(before), "n_after": len(after),
                    "sha256_before": hashlib.sha256(before.tobytes()).hexdigest(), "sha256_after": hashlib.sha256(after.tobytes()).hexdigest(),
                    "known_R": np.asarray(R_known, np.float64).tolist(), "known_t": np.asarray(t_known, np.float64).tolist(),
                    "cpu_slam": {"repetitions": int(nrep), "R": np.asarray(R, np.float64).tolist(), "t": np.asarray(t, np.float64).tolist(), "error": float(err), "seconds": round(dt, 1)},
                    "cpu_slam_vs_known": frob(R, t, R_known, t_known)})
        print("%-11s %6d points of %s: cpu-slam %2d repetitions, error %.4g, |d| to the known motion %.3g (%.1f s)" % (which, size, name, nrep, err, out[-1]["cpu_slam_vs_known"], dt), flush=True)
    with open(os.path.join(GOLD, "nicp_sets.json"), "w") as f:
        json.dump({"source": "GetSizesTestSet / GetPerformanceTestSet (testset.cpp:48-116), NICP, run by oracle/_ref; see oracle/make_golden_nicp_sets.py", "configs": out}, f, indent=1)
    print("wrote tests/golden/nicp_sets.json")


if __name__ == "__main__":
    main()
