"""CPU suite: the Fast-Gauss-Transform restatement (oracle/fgt_oracle.c) against the fixtures generated from the reference's own
CPU build (tests/golden/bunny_fgt*.{json,npz}, oracle/make_golden_fgt.py), against that build run live where it is present
(oracle/_ref), and the host-side monomial tables of the product library against the restatement."""
import numpy as np
import pytest

from conftest import frob


def small_clouds(seed, m, n):
    rng = np.random.default_rng(seed)
    y = rng.normal(size=(m, 3)).astype(np.float32) * 2
    x = (y[rng.integers(0, m, n)] + rng.normal(size=(n, 3)) * 0.3).astype(np.float32)
    return y, x


# ---------------------------------------------------------------------------------------------------------------------
# against the committed fixtures
# ---------------------------------------------------------------------------------------------------------------------
def test_bunny_kcenter_labels_and_means_bit_exact(oracle, golden, bunny):
    _, after = bunny
    e = golden.npz("bunny_fgt_estep.npz")
    xc, labels = oracle.fgt_kcenter(after, 117)
    assert np.array_equal(labels, e["kcenter117_labels"].astype(np.int32))
    assert np.array_equal(xc, e["kcenter117_xc"])
    assert labels[1] == 0 and set(np.unique(labels)) == set(range(117))      # the sweep starts from point 1 (fgt.cpp:162)


@pytest.mark.parametrize("name", ["init", "s006"])
def test_bunny_fgt_estep_bit_exact(oracle, golden, bunny, name):
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    e = golden.npz("bunny_fgt_estep.npz")
    c = g["esteps"][name]
    st = g["stride"]
    assert oracle.cpd_fgt_clusters(len(before), len(after), c["sigma2"], g["sigma2_init"]) == c["K"]
    assert oracle.cpd_fgt_ndi(c["sigma2"], g["weight"], len(before), len(after)) == c["ndi"]
    p1, pt1, px, L = oracle.cpd_estep_fgt(before, after, g["weight"], c["sigma2"], g["sigma2_init"])
    assert np.array_equal(p1[::st], e[name + "_p1"]) and np.array_equal(pt1[::st], e[name + "_pt1"])
    assert np.array_equal(px[::st], e[name + "_px"])
    assert L == c["L"]
    assert float(p1.astype(np.float64).sum()) == c["p1_sum"] and float(pt1.astype(np.float64).sum()) == c["pt1_sum"]


def test_bunny_truncated_estep_bit_exact(oracle, golden, bunny):
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    e = golden.npz("bunny_fgt_estep.npz")
    t = g["truncated"]
    p1, pt1, px, L = oracle.cpd_estep_truncated(before, after, g["constant"], t["sigma2"], t["truncate"])
    st = g["stride"]
    assert np.array_equal(p1[::st], e["trunc_p1"]) and np.array_equal(pt1[::st], e["trunc_pt1"]) and np.array_equal(px[::st], e["trunc_px"])
    assert L == t["L"]


def test_bunny_hybrid_full_run(oracle, golden, bunny):
    # the parser's default approximation type: FGT E-steps while sigma^2 > 0.015 sigma^2_init, truncated exact ones after
    before, after = bunny
    g = golden.json("bunny_fgt.json")["runs"]["hybrid"]
    R, t, it, err, trace = oracle.cpd_approx(before, after, oracle.APPROX_HYBRID, max_iterations=g["max_iterations"], trace_cap=64)
    assert it == g["iterations"] == 23
    assert frob(R, t, np.array(g["R"]), np.array(g["t"])) < 2e-5
    assert list(trace[:, 16]) == [1.0] * 18 + [0.0] * 5                       # 18 FGT E-steps, then 5 truncated ones
    assert err < 1e-3                                                         # the final sigma^2 is cancellation noise


@pytest.mark.parametrize("cap", [5, 17])
def test_bunny_full_mode_capped(oracle, golden, bunny, cap):
    # approximation "full" keeps using the FGT at bandwidths where p = 8 no longer converges: after ~18 iterations the
    # reference itself oscillates (sigma^2 jumps back up), so parity is anchored on runs capped before that
    before, after = bunny
    g = golden.json("bunny_fgt.json")["runs"]["full_cap%d" % cap]
    R, t, it, err = oracle.cpd_approx(before, after, oracle.APPROX_FULL, max_iterations=cap)
    assert it == g["iterations"] == cap
    # mid-trajectory the scale estimate carries the M-step's summation-order noise (Eigen's float sums vs fp64)
    assert frob(R, t, np.array(g["R"]), np.array(g["t"])) < 1e-4
    assert abs(err - g["error"]) < 2e-4 * g["error"]


# ---------------------------------------------------------------------------------------------------------------------
# against the reference build run live
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,n,K,p,sigma", [(0, 900, 50, 8, 2.5), (1, 300, 117, 8, 0.4), (2, 64, 7, 3, 1.0), (3, 40, 40, 5, 0.7),
                                             (4, 500, 1, 8, 3.0)])
def test_fgt_model_and_predict_match_reference(oracle, ref, seed, n, K, p, sigma):
    src, qry = small_clouds(seed, n, 2 * n + 3)
    w = np.random.default_rng(seed).random(n).astype(np.float32)
    xr, ar = ref.fgt_model(src, w, sigma, K, p)
    xo, ao = oracle.fgt_model(src, w, sigma, K, p)
    assert np.array_equal(xr, xo) and np.array_equal(ar, ao)
    for e in (10.0, 2.0):
        assert np.array_equal(ref.fgt_predict(qry, xr, ar, sigma, e, p), oracle.fgt_predict(qry, xo, ao, sigma, e, p))


def test_fgt_duplicate_points_leave_empty_cells_like_the_reference(oracle, ref):
    # more cells than distinct points: the surplus cells stay empty and their means are 0 * inf = NaN in the reference too
    base = np.random.default_rng(5).normal(size=(6, 3)).astype(np.float32)
    cloud = np.repeat(base, 5, axis=0)
    w = np.ones(len(cloud), np.float32)
    xr, ar = ref.fgt_model(cloud, w, 1.0, 10, 4)
    xo, ao = oracle.fgt_model(cloud, w, 1.0, 10, 4)
    assert np.array_equal(np.isnan(xr), np.isnan(xo)) and np.isnan(xr).any()
    assert np.array_equal(np.nan_to_num(xr, nan=7.0), np.nan_to_num(xo, nan=7.0))
    assert np.array_equal(np.nan_to_num(ar, nan=7.0), np.nan_to_num(ao, nan=7.0))


@pytest.mark.parametrize("seed,m,n,s2", [(0, 400, 500, 2.0), (1, 700, 300, 0.3), (2, 100, 100, 0.05)])
def test_fgt_and_truncated_esteps_match_reference(oracle, ref, seed, m, n, s2):
    y, x = small_clouds(seed, m, n)
    s2_init = 4.0
    a = ref.cpd_estep_fgt(y, x, 0.3, s2, s2_init)
    b = oracle.cpd_estep_fgt(y, x, 0.3, s2, s2_init)
    for u, v in zip(a[:3], b[:3]):
        assert np.array_equal(u, v)
    assert a[3] == b[3]
    c = oracle.cpd_constant(s2_init, 0.3, m, n)
    a = ref.cpd_estep_truncated(y, x, c, s2, 1e-3)
    b = oracle.cpd_estep_truncated(y, x, c, s2, 1e-3)
    for u, v in zip(a[:3], b[:3]):
        assert np.array_equal(u, v)
    assert a[3] == b[3]


@pytest.mark.parametrize("approx", [1, 2])
def test_small_approximate_runs_match_reference(oracle, ref, approx):
    rng = np.random.default_rng(11)
    b = rng.normal(size=(600, 3)).astype(np.float32) * 2
    ang = 0.3
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    a = (b[rng.permutation(600)] @ Rz.T + np.array([0.4, -0.2, 0.1])).astype(np.float32)
    cap = 12
    Rr, tr, itr, er = ref.cpd(b, a, 1e-3, 0.3, False, cap, 1e-3, approx)
    Ro, to, ito, eo = oracle.cpd_approx(b, a, approx, max_iterations=cap)
    assert itr == ito
    assert frob(Rr, tr, Ro, to) < 5e-5
    assert abs(er - eo) < 1e-3 * max(er, 1e-3)


# ---------------------------------------------------------------------------------------------------------------------
# the product library's host-side tables (no device needed)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("p", [1, 2, 3, 5, 8, 11, 16])
def test_monomial_tables_match_the_restatement(capi, oracle, p):
    exps, ck, slot = capi.fgt_tables(p)
    pd = oracle.fgt_pd(p)
    assert len(ck) == pd == (p + 2) * (p + 1) * p // 6
    assert np.array_equal(ck, oracle.fgt_ck(p))                                # ComputeC_k's step-by-step rounding
    assert exps.sum(axis=1).max() == p - 1 and len({tuple(e) for e in exps}) == pd
    assert sorted(slot) == list(range(pd))


def test_tables_reproduce_the_reference_transform(capi, oracle):
    # evaluate the transform from the tables alone (numpy, fp64): monomial order = the reference's coefficient order, and the
    # Horner traversal over `slot` gives the same polynomial
    p, K, sigma, e = 6, 9, 1.3, 10.0
    src, qry = small_clouds(3, 200, 50)
    w = np.random.default_rng(0).random(200).astype(np.float32)
    xc, ak = oracle.fgt_model(src, w, sigma, K, p)
    want = oracle.fgt_predict(qry, xc, ak, sigma, e, p)
    exps, ck, slot = capi.fgt_tables(p)
    dy = (qry[:, None, :].astype(np.float64) - xc[None, :, :]) / sigma          # [q, K, 3]
    near = (dy ** 2).sum(-1) <= e
    mono = np.prod(dy[:, :, None, :] ** exps[None, None, :, :], axis=-1)        # [q, K, pd]
    direct = (np.exp(-(dy ** 2).sum(-1))[:, :, None] * mono * ak[None]).sum(-1)
    assert np.allclose((direct * near).sum(1), want, rtol=2e-4, atol=1e-5)
    B = np.empty_like(ak)
    B[:, slot] = ak                                                             # what the model kernel stores
    got = np.zeros(len(qry))
    for q in range(len(qry)):
        for k in range(K):
            if not near[q, k]:
                continue
            x, y, z = dy[q, k]
            h, pa = 0, 0.0
            for a in range(p - 1, -1, -1):
                pb = 0.0
                for b in range(p - 1 - a, -1, -1):
                    pc = 0.0
                    for c in range(p - 1 - a - b, -1, -1):
                        pc = pc * z + B[k, h]
                        h += 1
                    pb = pb * y + pc
                pa = pa * x + pb
            got[q] += np.exp(-(x * x + y * y + z * z)) * pa
    assert np.allclose(got, want, rtol=2e-4, atol=1e-5)
