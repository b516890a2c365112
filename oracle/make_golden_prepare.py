"""TEST INFRASTRUCTURE ONLY.  Generates the input-stage fixture under tests/golden/ by running the reference's OWN
GetCloudsFromConfig stages (oracle/_ref: ref_clouds_from_config_full -- GetSubcloud, NormalizeCloud, std::shuffle, AddNoiseToCloud,
AddOutliersToCloud, GetTransformedCloud of source/common/common.cpp, on the reference's generators).  Run in the build container:

    python oracle/make_golden_prepare.py

Output
    tests/golden/bunny_prepare.npz   per case c0, c1: the options (JSON string), the random outcomes the reference drew in the
                                     form mi_prepare_cloud takes them (index vectors from the reference's generator, unit draws
                                     from the C library's rand()), and the reference's two prepared clouds.
The raw cloud is the first rows of the committed `before` bunny cloud (any cloud serves as LoadCloud's output).
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

ROT = [[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]]
CASES = [
    # everything on: both subclouds, normalisation, noise on both, outliers on both
    dict(raw_rows=9000, seed=666, spread=10.0, resize_before=3000, resize_after=3500, noise_before=[0.25, 0.05],
         noise_after=[0.1, 0.02], outliers_before=13, outliers_after=29, R=ROT, t=[1.0, 2.0, 3.0]),
    # no normalisation, "resize" beyond the cloud (GetSubcloud returns it untouched and draws nothing), every point of `after` noisy
    dict(raw_rows=2500, seed=12345, spread=None, resize_before=None, resize_after=4000, noise_before=None, noise_after=[1.0, 0.1],
         outliers_before=0, outliers_after=5, R=ROT, t=[-0.5, 0.25, 8.0]),
]


def main():
    raw_all = np.load(os.path.join(GOLD, "bunny_clouds.npz"))["before"]
    out = {}
    for k, case in enumerate(CASES):
        raw = raw_all[:case["raw_rows"]]
        kw = dict(resize_before=case["resize_before"], resize_after=case["resize_after"],
                  noise_before=None if case["noise_before"] is None else tuple(case["noise_before"]),
                  noise_after=None if case["noise_after"] is None else tuple(case["noise_after"]),
                  outliers_before=case["outliers_before"], outliers_after=case["outliers_after"])
        before, after = ref.clouds_from_config_full(raw, None, case["seed"], np.array(case["R"], np.float32),
                                                    np.array(case["t"], np.float32), spread=case["spread"], **kw)
        db, da = ref.config_draws(len(raw), len(raw), case["seed"], **kw)
        out["c%d_options" % k] = np.array(json.dumps(case))
        out["c%d_before" % k], out["c%d_after" % k] = before, after
        for side, d in (("b", db), ("a", da)):
            for name, v in d.items():
                if v is not None:
                    small = v.dtype == np.int32 and (v.size == 0 or v.max() < 65536)
                    out["c%d_%s_%s" % (k, side, name)] = v.astype(np.uint16) if small else v
    np.savez_compressed(os.path.join(GOLD, "bunny_prepare.npz"), **out)
    print("wrote bunny_prepare.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
