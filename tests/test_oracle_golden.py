"""CPU suite: the plain-C restatement (oracle/slam_oracle.c) against the golden vectors the REFERENCE's own cpu-slam code
produced (tests/golden/, made by oracle/make_golden.py from oracle/_ref).  This is what pins the oracle.

Tolerances: indices / iteration counts bit-exact; float results to the stated bounds -- the only arithmetic the
restatement does not reproduce bit-for-bit is Eigen's blocked float GEMM/GEMV/redux summation order (machine-dependent),
restated with double accumulators.
"""
import numpy as np
import pytest

from conftest import frob


def test_correspondences_known_answer(oracle):
    # CorrespondencesTest, source/cuda-slam/cudacommon.cu:291-317: input[i] = (i,i,i), output[99-i] = (i,i,i) => idx[i] = 99-i
    n = 100
    src = np.repeat(np.arange(n, dtype=np.float32)[:, None], 3, axis=1)
    tgt = src[::-1].copy()
    idx, d2 = oracle.nn_search(src, tgt)
    assert np.array_equal(idx, n - 1 - np.arange(n))
    assert np.all(d2 == 0)


def test_nn_first_index_wins_ties(oracle):
    # strict '<' with ascending j (common.cpp:454): duplicated targets resolve to the lowest index
    rng = np.random.default_rng(1)
    tgt = rng.uniform(-1, 1, (50, 3)).astype(np.float32)
    tgt = np.concatenate([tgt, tgt, tgt])           # every point three times
    src = tgt[50:100] + np.float32(1e-3)
    idx, _ = oracle.nn_search(src, tgt, threads=3)
    assert np.array_equal(idx, np.arange(50))


def test_bunny_iter0_correspondences_bit_exact(oracle, golden, bunny):
    before, after = bunny
    g = golden.npz("bunny_icp_iter0.npz")
    idx, d2 = oracle.nn_search(before, after)
    ib = oracle.filter_pairs(d2, 400.0)
    assert np.array_equal(ib, g["idx_before"])
    assert np.array_equal(idx[ib], g["idx_after"])


def test_bunny_iter0_kabsch(oracle, golden, bunny):
    before, after = bunny
    g = golden.npz("bunny_icp_iter0.npz")
    R, t = oracle.least_squares_svd(before[g["idx_before"]], after[g["idx_after"]])
    assert np.abs(R - g["R0"]).max() < 2e-6
    assert np.abs(t - g["t0"]).max() < 2e-6


def test_jacobi_svd3_reconstructs(oracle):
    rng = np.random.default_rng(3)
    for _ in range(50):
        A = rng.normal(size=(3, 3)).astype(np.float32) * np.float32(10 ** rng.uniform(-3, 3))
        U, S, V = oracle.jacobi_svd3(A)
        assert np.all(S[:-1] >= S[1:]) and np.all(S >= 0)
        assert np.abs(U @ np.diag(S) @ V.T - A).max() <= 4e-6 * np.abs(A).max()
        assert np.abs(U.T @ U - np.eye(3)).max() < 1e-5
        assert np.abs(V.T @ V - np.eye(3)).max() < 1e-5


def test_bunny_icp_full_run(oracle, golden, bunny):
    # cfg 1: BasicICP::GetBasicICPTransformationMatrix on config/default.json -- iteration count exact, R|t to 1e-5
    before, after = bunny
    g = golden.json("bunny_icp.json")
    p = g["params"]
    R, t, it, err = oracle.icp(before, after, p["eps"], p["max_distance_squared"], p["max_iterations"])
    assert it == g["iterations"] == 39
    assert frob(R, t, g["R"], g["t"]) < 1e-5
    assert abs(err - g["error"]) < 1e-6


@pytest.mark.parametrize("k", [1, 2, 3, 5, 10, 20])
def test_bunny_icp_capped(oracle, golden, bunny, k):
    before, after = bunny
    g = golden.json("bunny_icp.json")
    c = g["capped"][str(k)]
    R, t, it, err = oracle.icp(before, after, 1e-3, 400.0, k)
    assert it == c["iterations"]
    assert frob(R, t, c["R"], c["t"]) < 1e-5
    assert abs(err - c["error"]) < 1e-5 * max(1.0, c["error"])


def test_synth2k_icp(oracle, golden):
    z = golden.npz("synth2k_clouds.npz")
    g = golden.json("synth2k_icp.json")
    gi = golden.npz("synth2k_icp_iter0.npz")
    idx, d2 = oracle.nn_search(z["before"], z["after"])
    ib = oracle.filter_pairs(d2, 1000.0)
    assert np.array_equal(ib, gi["idx_before"]) and np.array_equal(idx[ib], gi["idx_after"])
    R, t, it, err = oracle.icp(z["before"], z["after"], 1e-3, 1000.0, 60)
    assert it == g["iterations"]
    assert frob(R, t, g["R"], g["t"]) < 1e-5


def test_icp_driver_modes(oracle, golden):
    # exact composition + no filter + abort-on-increase (the cuda-slam driver rules) reach the same fixed point on a
    # clean synthetic problem; with max_iterations = 0 nothing runs (identity, error 1e5)
    z = golden.npz("synth2k_clouds.npz")
    R, t, it, err = oracle.icp(z["before"], z["after"], 1e-3, 1000.0, 60, compose_mode=oracle.COMPOSE_EXACT,
                               filter_pairs=False, abort_on_increase=True, dist_mode=oracle.DIST_FMA)
    assert err < 1e-3
    assert frob(R, t, z["R_true"], z["t_true"]) < 5e-2
    R0, t0, it0, err0 = oracle.icp(z["before"], z["after"], max_iterations=0)
    assert it0 == 0 and err0 == pytest.approx(1e5) and np.array_equal(R0, np.eye(3)) and np.all(t0 == 0)


def test_bunny_cpd_init_and_estep_bit_exact(oracle, golden, bunny):
    before, after = bunny
    g = golden.json("bunny_cpd.json")
    s2 = oracle.cpd_sigma_squared(before, after)
    assert s2 == g["sigma2_init"]                    # the reference's saturated sequential fp32 sum, bit for bit
    c = oracle.cpd_constant(s2, 0.3, len(before), len(after))
    assert c == g["constant"]
    p1, pt1, px, L = oracle.cpd_estep(before, after, c, s2)
    e = golden.npz("bunny_cpd_estep0.npz")
    assert np.array_equal(p1, e["p1"]) and np.array_equal(pt1, e["pt1"]) and np.array_equal(px, e["px"])
    assert L == g["L0"]


@pytest.mark.parametrize("const_scale,key", [(False, "mstep0_scale_free"), (True, "mstep0_const_scale")])
def test_bunny_cpd_mstep(oracle, golden, bunny, const_scale, key):
    before, after = bunny
    g = golden.json("bunny_cpd.json")[key]
    e = golden.npz("bunny_cpd_estep0.npz")
    R, t, s, s2 = oracle.cpd_mstep(before, after, e["p1"], e["pt1"], e["px"], const_scale)
    assert np.abs(R - np.array(g["R"])).max() < 5e-6
    assert np.abs(t - np.array(g["t"])).max() < 2e-5
    assert abs(s - g["scale"]) < 2e-5 * g["scale"]
    assert abs(s2 - g["sigma2"]) < 5e-5 * g["sigma2"]


def test_bunny_cpd_full_run(oracle, golden, bunny):
    # cfg 4: GetRigidCPDTransformationMatrix, approximation none, parser defaults (weight .3, const-scale false)
    before, after = bunny
    g = golden.json("bunny_cpd.json")
    f = g["final_scale_free"]
    sR, t, it, err = oracle.cpd(before, after, 1e-3, 0.3, False, 50, 1e-3)
    assert it == f["iterations"]
    assert frob(sR, t, f["sR"], f["t"]) < 1e-4
    # The final sigma^2 is |sigmaSubtrahend - scale*scaleNumerator| / (3 Np): a difference of two ~1e5-sized fp32 numbers
    # that agree to ~7 digits at convergence, i.e. rounding noise of a few 1e-5 in the reference itself.
    assert abs(err - f["error"]) < 1e-4
