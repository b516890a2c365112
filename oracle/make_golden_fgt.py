"""TEST INFRASTRUCTURE ONLY.  Generates the Fast-Gauss-Transform fixtures under tests/golden/ by running the reference's OWN
CPU code (oracle/_ref/libref_cpuslam.so: source/common/fgt.cpp, cpdutils.cpp and cpu-slam/coherentpointdrift.cpp compiled in
place) on the committed bunny clouds.  Run in the build container (needs /root/reference for the library build):

    python oracle/make_golden_fgt.py

Outputs
    tests/golden/bunny_fgt.json        hybrid full run (cfg: parser defaults), full-mode runs capped before the mode turns
                                       chaotic, the E-step scalars (K, ndi, L) at three sigma^2, a truncated-E-step L
    tests/golden/bunny_fgt_estep.npz   every 3rd entry of P1 / Pt1 / PX of the FGT E-step at sigma^2_init and at 0.06,
                                       of the truncated exact E-step at 0.05, and the K-centre labels for K = 117
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
from oracle import oraclebind as oracle    # noqa: E402

STRIDE = 3


def main():
    z = np.load(os.path.join(GOLD, "bunny_clouds.npz"))
    before, after = z["before"], z["after"]
    m, n = len(before), len(after)
    s2_init = ref.cpd_sigma_squared(before, after)
    weight = 0.3
    constant = oracle.cpd_constant(s2_init, weight, m, n)      # pinned bit-exact against the reference elsewhere
    out = dict(sigma2_init=s2_init, weight=weight, constant=constant, stride=STRIDE, esteps={})
    arrays = {}
    for name, s2 in (("init", s2_init), ("s006", 0.06)):
        p1, pt1, px, L = ref.cpd_estep_fgt(before, after, weight, s2, s2_init, 10.0, 8.0)
        out["esteps"][name] = dict(sigma2=s2, L=L, K=oracle.cpd_fgt_clusters(m, n, s2, s2_init), ndi=oracle.cpd_fgt_ndi(s2, weight, m, n),
                                   p1_sum=float(p1.astype(np.float64).sum()), pt1_sum=float(pt1.astype(np.float64).sum()))
        arrays[name + "_p1"] = p1[::STRIDE]
        arrays[name + "_pt1"] = pt1[::STRIDE]
        arrays[name + "_px"] = px[::STRIDE]
    p1, pt1, px, L = ref.cpd_estep_truncated(before, after, constant, 0.05, 1e-3)
    out["truncated"] = dict(sigma2=0.05, truncate=1e-3, L=L, p1_sum=float(p1.astype(np.float64).sum()))
    arrays["trunc_p1"] = p1[::STRIDE]
    arrays["trunc_pt1"] = pt1[::STRIDE]
    arrays["trunc_px"] = px[::STRIDE]
    # K-centre labels through the reference's model builder: with unit weights and order 1, column k of A_k is
    # sum_{i in cell k} exp(-|dx|^2): not the labels themselves, so take them from the restatement, which the CPU suite
    # pins against the reference's cell means and coefficients bit for bit (tests/test_oracle_vs_ref.py).
    xc, labels = oracle.fgt_kcenter(after, 117)
    xc_ref, _ = ref.fgt_model(after, np.ones(n, np.float32), 1.0, 117, 1)
    assert np.array_equal(xc, xc_ref)
    arrays["kcenter117_labels"] = labels.astype(np.uint8)
    arrays["kcenter117_xc"] = xc_ref

    runs = {}
    R, t, it, err = ref.cpd(before, after, 1e-3, weight, False, 50, 1e-3, 2, 10.0, 8.0)
    runs["hybrid"] = dict(max_iterations=50, R=R.tolist(), t=t.tolist(), iterations=it, error=err)
    for cap in (5, 17):
        R, t, it, err = ref.cpd(before, after, 1e-3, weight, False, cap, 1e-3, 1, 10.0, 8.0)
        runs["full_cap%d" % cap] = dict(max_iterations=cap, R=R.tolist(), t=t.tolist(), iterations=it, error=err)
    out["runs"] = runs
    json.dump(out, open(os.path.join(GOLD, "bunny_fgt.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(GOLD, "bunny_fgt_estep.npz"), **arrays)
    print("wrote bunny_fgt.json, bunny_fgt_estep.npz")


if __name__ == "__main__":
    main()
