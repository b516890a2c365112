"""CPU suite: the plain-C restatement against the reference's OWN cpu-slam code run live (oracle/_ref/libref_cpuslam.so,
compiled by oracle/Makefile from /root/reference; shipped prebuilt to the GPU box).  Skipped where that library is absent.
Sizes are small: the whole file runs in seconds.
"""
import numpy as np
import pytest


def clouds(seed, n, m, dup=False):
    rng = np.random.default_rng(seed)
    a = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    b = rng.uniform(-5, 5, (m, 3)).astype(np.float32)
    if dup:
        b[m // 2:] = b[: m - m // 2]          # duplicated targets: ties everywhere
        a[: n // 4] = b[: n // 4]             # exact hits
    return a, b


@pytest.mark.parametrize("seed,n,m,dup", [(0, 257, 1000, False), (1, 1000, 333, True), (2, 1, 5, False), (3, 64, 1, False)])
@pytest.mark.parametrize("parallel", [True, False])
def test_correspondences_match_reference(oracle, ref, seed, n, m, dup, parallel):
    src, tgt = clouds(seed, n, m, dup)
    for maxd in (1000.0, 2.0):
        ib_r, ia_r = ref.corresponding_points(src, tgt, maxd, parallel)
        idx, d2 = oracle.nn_search(src, tgt, threads=0 if parallel else 1)
        ib = oracle.filter_pairs(d2, maxd)
        assert np.array_equal(ib, ib_r)
        assert np.array_equal(idx[ib], ia_r)


@pytest.mark.parametrize("seed", range(5))
def test_least_squares_svd_matches_reference(oracle, ref, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 400))
    b = rng.normal(size=(n, 3)).astype(np.float32) * 3
    ang = rng.uniform(0, np.pi)
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    a = (b @ Rm.T + rng.normal(size=3) * 4 + rng.normal(size=(n, 3)) * 0.05).astype(np.float32)
    Rr, tr = ref.least_squares_svd(b, a)
    Ro, to = oracle.least_squares_svd(b, a)
    assert np.abs(Rr - Ro).max() < 5e-6
    assert np.abs(tr - to).max() < 5e-5


def test_least_squares_svd_reflection_case(oracle, ref):
    # planar, mirrored configuration: det(U V^T) = -1 branch of common.cpp:541-545
    rng = np.random.default_rng(11)
    b = rng.normal(size=(50, 3)).astype(np.float32)
    b[:, 2] = 0
    a = b.copy(); a[:, 0] = -a[:, 0]
    Rr, tr = ref.least_squares_svd(b, a)
    Ro, to = oracle.least_squares_svd(b, a)
    assert np.abs(Rr - Ro).max() < 1e-5 and np.abs(tr - to).max() < 1e-5
    assert np.linalg.det(Ro.astype(np.float64)) > 0.99


def test_transform_and_mse_bit_exact(oracle, ref):
    src, tgt = clouds(5, 500, 400)
    R = np.array([[0.36, 0.47, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32)
    t = np.array([1.0, -2.0, 0.5], np.float32)
    assert np.array_equal(ref.transform_cloud(src, R, t), oracle.transform_cloud(src, R, t))
    ib = np.arange(500, dtype=np.int32)
    ia = (np.arange(500, dtype=np.int32) * 7) % 400
    assert ref.mse_indexed(src, tgt, ib, ia) == oracle.mse_indexed(src, tgt, ib, ia)


@pytest.mark.parametrize("seed,n,rot,trans", [(7, 600, 0.2, 1.0), (8, 900, 0.4, 0.5)])
def test_icp_full_run_matches_reference(oracle, ref, seed, n, rot, trans):
    rng = np.random.default_rng(seed)
    b = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rm = np.eye(3) + np.sin(rot) * K + (1 - np.cos(rot)) * K @ K
    a = (b[rng.permutation(n)] @ Rm.T + trans).astype(np.float32)
    Rr, tr, itr, er = ref.icp(b, a, 1e-3, 1000.0, 40, True)
    Ro, to, ito, eo = oracle.icp(b, a, 1e-3, 1000.0, 40)
    assert itr == ito
    assert np.sqrt(((Rr - Ro) ** 2).sum() + ((tr - to) ** 2).sum()) < 1e-5
    assert abs(er - eo) <= 1e-5 * max(er, 1e-3)


def test_cpd_pieces_match_reference(oracle, ref):
    src, tgt = clouds(9, 300, 350)
    tgt = (src[:350 % 300 + 250] if False else tgt)
    s2r, s2o = ref.cpd_sigma_squared(src, tgt), oracle.cpd_sigma_squared(src, tgt)
    assert s2r == s2o
    c = oracle.cpd_constant(s2o, 0.3, len(src), len(tgt))
    for sigma2 in (s2o, 0.5, 0.01):
        r = ref.cpd_estep(src, tgt, c, sigma2)
        o = oracle.cpd_estep(src, tgt, c, sigma2)
        for x, y in zip(r[:3], o[:3]):
            assert np.array_equal(x, y)
        assert r[3] == o[3]
    p1, pt1, px, _ = o
    for cs in (False, True):
        Rr, tr, sr, s2nr = ref.cpd_mstep(src, tgt, p1, pt1, px, cs)
        Ro, to, so, s2no = oracle.cpd_mstep(src, tgt, p1, pt1, px, cs)
        assert np.abs(Rr - Ro).max() < 1e-5 and np.abs(tr - to).max() < 1e-4
        # sigma^2 is a cancelling difference of ~Np*|x|^2-sized fp32 terms: absolute noise floor of a few 1e-5
        assert abs(sr - so) < 1e-4 * abs(sr) and abs(s2nr - s2no) < 1e-3 * abs(s2nr) + 5e-5


def test_cpd_full_run_matches_reference(oracle, ref):
    rng = np.random.default_rng(21)
    b = rng.uniform(-5, 5, (400, 3)).astype(np.float32)
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    Rm = np.eye(3) + np.sin(0.3) * K + (1 - np.cos(0.3)) * K @ K
    a = (b[rng.permutation(400)] @ Rm.T + 1.0).astype(np.float32)
    for cs in (False, True):
        Rr, tr, itr, er = ref.cpd(b, a, 1e-3, 0.3, cs, 40, 1e-3, 0)
        Ro, to, ito, eo = oracle.cpd(b, a, 1e-3, 0.3, cs, 40, 1e-3)
        assert itr == ito
        assert np.sqrt(((Rr - Ro) ** 2).sum() + ((tr - to) ** 2).sum()) < 1e-4
    # the reference's own quirk (coherentpointdrift.cpp:106): max_iterations = -1 runs nothing
    Rr, tr, itr, er = ref.cpd(b, a, 1e-3, 0.3, False, -1, 1e-3, 0)
    Ro, to, ito, eo = oracle.cpd(b, a, 1e-3, 0.3, False, -1, 1e-3)
    assert itr == ito == 0 and np.array_equal(Rr, Ro) and np.array_equal(Rr, np.eye(3, dtype=np.float32))
