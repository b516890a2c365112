#!/usr/bin/env python3
"""Developer tool: one bunny CPD run in hybrid mode (18 FGT E-steps + 5 truncated exact ones), the command profiled under
rocprofv3 --kernel-trace --stats for the K9 kernel breakdown in profiles/."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402


def main():
    capi = load_package().capi
    z = np.load(os.path.join(ROOT, "tests", "golden", "bunny_clouds.npz"))
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "bunny_fgt.json")))
    ctx = capi.Context(0)
    p = capi.cpd_params(max_iterations=50, sigma2_init=gold["sigma2_init"], approximation=capi.CPD_APPROX_HYBRID)
    for _ in range(3):
        sR, t, scale, it, err = ctx.cpd_register(z["before"], z["after"], p)
    print(json.dumps({"iterations": it, "sigma2": err, "t": [float(x) for x in t]}))
    ctx.close()


if __name__ == "__main__":
    main()
