"""GPU suite: the Fast-Gauss-Transform E-step and the full / hybrid CPD drivers through the C ABI, against the plain-C
restatement (oracle/fgt_oracle.c, itself bit-identical to the reference's CPU build on these paths) and the fixtures generated
from that build (tests/golden/bunny_fgt*).  Integer work (the K-centre labels) must match bit for bit; the transforms differ
from cpu-slam by expf's last bit and the polynomial evaluation order only -- tolerances are written next to each check."""
import numpy as np
import pytest

from conftest import check_measured, frob

pytestmark = pytest.mark.gpu


def cloud(seed, n, spread=2.0):
    return (np.random.default_rng(seed).normal(size=(n, 3)) * spread).astype(np.float32)


def pair(seed, m, n):
    rng = np.random.default_rng(seed)
    y = (rng.normal(size=(m, 3)) * 2).astype(np.float32)
    x = (y[rng.integers(0, m, n)] + rng.normal(size=(n, 3)) * 0.3).astype(np.float32)
    return y, x


def rel(a, b):
    return float(np.abs(a.astype(np.float64) - b).max() / max(np.abs(b).max(), 1e-30))


# ---------------------------------------------------------------------------------------------------------------------
# K-centre clustering: labels and cell means bit for bit
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,n,K", [(0, 2, 1), (1, 2, 2), (2, 300, 50), (3, 4096, 117), (4, 4097, 64), (5, 14904, 117),
                                      (6, 16384, 51), (7, 16385, 33), (8, 50000, 70), (9, 1000, 1000), (10, 65536, 20), (11, 65537, 37),
                                      (12, 300000, 51), (13, 5000, 3000), (14, 20000, 9000)])     # (the last two: member lists beyond 1 024 / 8 192 cells)
def test_kcenter_matches_oracle_bit_for_bit(ctx, oracle, seed, n, K):
    c = cloud(seed, n)
    xc_o, lab_o = oracle.fgt_kcenter(c, K)
    xc, lab = ctx.fgt_kcenter(c, K)
    assert np.array_equal(lab, lab_o)
    assert np.array_equal(xc, xc_o)


def test_kcenter_ties_and_duplicates(ctx, oracle):
    # a lattice (many equal distances: the FIRST maximum must win) with every point duplicated (strict < keeps the old label)
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    c = np.concatenate([g, g])[np.random.default_rng(0).permutation(1024)]
    for K in (2, 9, 64, 200):
        xc_o, lab_o = oracle.fgt_kcenter(c, K)
        xc, lab = ctx.fgt_kcenter(c, K)
        assert np.array_equal(lab, lab_o)
        assert np.array_equal(xc, xc_o)
    # more cells than distinct points: empty cells, NaN means, exactly as the reference
    few = np.repeat(cloud(3, 6), 5, axis=0)
    xc_o, lab_o = oracle.fgt_kcenter(few, 10)
    xc, lab = ctx.fgt_kcenter(few, 10)
    assert np.array_equal(lab, lab_o) and np.array_equal(np.isnan(xc), np.isnan(xc_o)) and np.isnan(xc).any()
    assert np.array_equal(np.nan_to_num(xc, nan=7.0), np.nan_to_num(xc_o, nan=7.0))


# ---------------------------------------------------------------------------------------------------------------------
# the sweep replayed from a guess (round 4): whatever the guess, the result is the step-by-step sweep's -- bit for bit
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,n,K", [(2, 300, 50), (3, 4096, 117), (5, 14904, 150), (7, 16385, 33), (8, 50000, 70), (11, 65537, 37),
                                      (12, 300000, 51)])
def test_guided_kcenter_is_the_sweep_whatever_the_guess(ctx, oracle, seed, n, K):
    c = cloud(seed, n)
    xc_o, lab_o = oracle.fgt_kcenter(c, K)
    xc0, lab0, picked0, v0 = ctx.fgt_kcenter_guided(c, K, np.zeros(0, np.int32))          # no guess: the plain sweep, and its choices
    assert v0 == -1 and np.array_equal(lab0, lab_o) and np.array_equal(xc0, xc_o)
    assert picked0[0] == 1 and len(set(picked0.tolist())) == K and np.array_equal(lab0[picked0], np.arange(K))   # centre k is a point of cell k
    rng = np.random.default_rng(seed)
    wrong_at = int(rng.integers(1, K - 1))
    bad = picked0.copy()
    bad[wrong_at] = (bad[wrong_at] + 1) % n
    shifted = picked0.copy()
    shifted[0] = 0                                                                        # centre 0 is point 1, always
    cases = [(picked0, K),                                                                # the whole sweep guessed right
             (picked0[: K // 2], K // 2),                                                 # half of it: the rest is swept
             (np.concatenate([picked0, picked0]), K),                                     # longer than K: the surplus is ignored
             (bad, wrong_at),                                                             # right up to wrong_at, then anything
             (shifted, 0),
             (rng.integers(0, n, K).astype(np.int32), None)]                              # noise
    for guess, want_verified in cases:
        xc, lab, picked, verified = ctx.fgt_kcenter_guided(c, K, guess)
        assert np.array_equal(lab, lab_o) and np.array_equal(xc, xc_o) and np.array_equal(picked, picked0)
        if len(guess) < 2:
            assert verified == -1
        elif want_verified is not None:
            assert verified == want_verified, (verified, want_verified)
        else:
            assert 0 <= verified <= K


@pytest.mark.parametrize("mode", ["0", "2"])
def test_cooperative_sweep_is_the_one_workgroup_sweep(capi, oracle, monkeypatch, mode):
    # Round 5: beyond 16 384 points a sweep of 16 or more steps runs on several workgroups in ONE cooperative launch (fgt_kcenter_coop_kernel: every
    # workgroup keeps its share in registers, a step's arg-max goes through memory behind a ticket barrier) -- the default, which the tests above run.
    # MISLAM_FGT_COOP_SWEEP=2 sends EVERY sweep of such a cloud there (also the few steps behind a replay: start read from the device), =0 none
    # (rounds 1-4: one workgroup with its distances in memory, two launches per centre beyond 65 536 points): the oracle's labels and means bit for bit
    # either way, plain and guided by right / cut / corrupted guesses, 1 / 2 / 4 / 16 points per lane, a last workgroup that is nearly empty
    monkeypatch.setenv("MISLAM_FGT_COOP_SWEEP", mode)
    with capi.Context(0) as c2:
        for seed, n, K in ((7, 16385, 33), (8, 50000, 70), (11, 65537, 37), (15, 131073, 20), (12, 300000, 51), (16, 1000001, 18)):
            c = cloud(seed, n)
            xc_o, lab_o = oracle.fgt_kcenter(c, K)
            xc, lab, picked, v = c2.fgt_kcenter_guided(c, K, np.zeros(0, np.int32))
            assert v == -1 and np.array_equal(lab, lab_o) and np.array_equal(xc, xc_o), (n, K)
            bad = picked.copy()
            bad[K // 2] = (bad[K // 2] + 1) % n
            for guess in (picked, picked[: K // 3], bad):
                xg, lg, pg, vg = c2.fgt_kcenter_guided(c, K, guess)
                assert np.array_equal(lg, lab_o) and np.array_equal(xg, xc_o) and np.array_equal(pg, picked), (n, K, len(guess))


def test_guided_kcenter_on_ties_and_under_similarity_transforms(ctx, oracle):
    # a lattice with duplicates: equal distances everywhere, the FIRST maximum must win in the replay as in the sweep
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    c = np.concatenate([g, g])[np.random.default_rng(0).permutation(1024)]
    for K in (9, 64, 200):
        xc_o, lab_o = oracle.fgt_kcenter(c, K)
        _, _, picked0, _ = ctx.fgt_kcenter_guided(c, K, np.zeros(0, np.int32))
        xc, lab, picked, verified = ctx.fgt_kcenter_guided(c, K, picked0)
        assert verified == K and np.array_equal(lab, lab_o) and np.array_equal(xc, xc_o) and np.array_equal(picked, picked0)
    # the use CPD makes of it: the same cloud rotated, scaled and shifted, guessed from the untransformed cloud's sweep.  On a lattice the
    # ties break differently once rounding enters (the guess fails early), on a scan-like cloud they hardly ever do -- either way the
    # labels are the oracle's for the TRANSFORMED cloud
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    for base, K in ((c, 64), (cloud(21, 9000), 120)):
        _, _, picked0, _ = ctx.fgt_kcenter_guided(base, K, np.zeros(0, np.int32))
        moved = (np.float32(1.1) * (base @ R.T) + np.array([0.4, -0.2, 0.1], np.float32)).astype(np.float32)
        xc_o, lab_o = oracle.fgt_kcenter(moved, K)
        xc, lab, picked, verified = ctx.fgt_kcenter_guided(moved, K, picked0)
        assert np.array_equal(lab, lab_o) and np.array_equal(np.nan_to_num(xc, nan=7.0), np.nan_to_num(xc_o, nan=7.0))
        assert 0 <= verified <= K
        first_diff = next((i for i in range(K) if picked[i] != picked0[i]), K)
        assert verified == first_diff                                                     # exactly the steps the two sweeps share


def test_guided_kcenter_beyond_the_replay_limit(ctx, oracle):
    # a replay stages at most 4 000 centres in LDS (FGT_REPLAY_MAX_CENTRES): a longer guess is replayed up to there, the rest is swept
    c = cloud(31, 12000)
    K = 4500
    xc_o, lab_o = oracle.fgt_kcenter(c, K)
    _, _, picked0, _ = ctx.fgt_kcenter_guided(c, K, np.zeros(0, np.int32))
    xc, lab, picked, verified = ctx.fgt_kcenter_guided(c, K, picked0)
    assert verified == 4000 and np.array_equal(lab, lab_o) and np.array_equal(xc, xc_o) and np.array_equal(picked, picked0)


def test_guided_kcenter_rejects_bad_guesses(ctx, capi):
    c = cloud(0, 100)
    with pytest.raises(capi.MiSlamError):
        ctx.fgt_kcenter_guided(c, 10, np.array([1, 100], np.int32))                       # not a point of the cloud
    with pytest.raises(capi.MiSlamError):
        ctx.fgt_kcenter_guided(c, 10, np.array([1, -1], np.int32))


def test_replayed_clustering_changes_nothing(ctx, capi, golden, bunny, monkeypatch):
    # every E-step but the first replays the moving cloud's previous sweep instead of sweeping step by step; switched off
    # (MISLAM_FGT_REPLAY=0, read at context creation) the runs must give the same bits
    monkeypatch.setenv("MISLAM_FGT_REPLAY", "0")
    with capi.Context(0) as plain:
        before, after = bunny
        g = golden.json("bunny_fgt.json")
        for approx, cap in ((capi.CPD_APPROX_HYBRID, 50), (capi.CPD_APPROX_FULL, 24)):
            p = capi.cpd_params(max_iterations=cap, sigma2_init=g["sigma2_init"], approximation=approx)
            a = ctx.cpd_register(before, after, p)
            b = plain.cpd_register(before, after, p)
            assert a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[4] == b[4]
        # a lattice against its rotated copy: guesses that fail early, every E-step
        lat = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(12), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
        ang = 0.2
        R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
        moved = (lat @ R.T + np.float32(0.3)).astype(np.float32)
        p = capi.cpd_params(max_iterations=12, tolerance=0.0, approximation=capi.CPD_APPROX_FULL)
        a = ctx.cpd_register(lat, moved, p)
        b = plain.cpd_register(lat, moved, p)
        assert a[3] == b[3] == 12 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[4] == b[4]


def test_bunny_kcenter_golden(ctx, golden, bunny):
    _, after = bunny
    e = golden.npz("bunny_fgt_estep.npz")
    xc, lab = ctx.fgt_kcenter(after, 117)
    assert np.array_equal(lab, e["kcenter117_labels"].astype(np.int32)) and np.array_equal(xc, e["kcenter117_xc"])


def test_kcenter_rejects_bad_arguments(ctx, capi):
    with pytest.raises(capi.MiSlamError):
        ctx.fgt_kcenter(cloud(0, 1), 1)              # the sweep starts from point 1: needs two points
    with pytest.raises(capi.MiSlamError):
        ctx.fgt_kcenter(cloud(0, 10), 0)
    with pytest.raises(capi.MiSlamError):
        ctx.fgt_kcenter(cloud(0, 10), 1 << 16)


# ---------------------------------------------------------------------------------------------------------------------
# E-steps
# ---------------------------------------------------------------------------------------------------------------------
# Tolerance: P1/Pt1/PX are sums of products of O(1e2) fp32 terms; against the reference's own evaluation order the Horner
# form agrees to a few 1e-6 of the largest entry, 1e-5 leaves headroom.
ESTEP_TOL = 1e-5


@pytest.mark.parametrize("seed,m,n,s2,order", [(0, 400, 500, 2.0, 8), (1, 700, 300, 0.3, 8), (2, 100, 100, 0.05, 8), (3, 2, 2, 1.0, 8),
                                               (4, 3000, 2500, 0.8, 5), (5, 513, 255, 1.5, 1), (6, 5000, 6000, 0.2, 11)])
def test_fgt_estep_matches_oracle(ctx, oracle, seed, m, n, s2, order):
    y, x = pair(seed, m, n)
    s2_init = 4.0
    want = oracle.cpd_estep_fgt(y, x, 0.3, s2, s2_init, 10.0, float(order))
    got = ctx.cpd_estep_fgt(y, x, 0.3, s2, s2_init, 10.0, order)
    for g, w in zip(got[:3], want[:3]):
        assert rel(g, w) < ESTEP_TOL
    assert abs(got[3] - want[3]) < 2e-6 * abs(want[3]) + 1e-3


def test_big_cells_split_over_workgroups(ctx, capi, oracle, monkeypatch):
    # Round 5: with thousands of members per cell the model build splits a cell's member list over several workgroups (partial sums added in a fixed
    # order) and takes the cell means in a launch of its own (fgt_centers_big_kernel: the same sequential sums).  Against MISLAM_FGT_MODEL_SPLITS=0 (one
    # workgroup per cell, rounds 1-4): the same arrays up to the summation order of a cell's coefficients; against the oracle: the usual bar.
    y, x = pair(21, 150000, 120000)
    s2, s2_init = 2.0, 4.0
    got = ctx.cpd_estep_fgt(y, x, 0.3, s2, s2_init, 10.0, 8)
    monkeypatch.setenv("MISLAM_FGT_MODEL_SPLITS", "0")
    with capi.Context(0) as one:
        ref = one.cpd_estep_fgt(y, x, 0.3, s2, s2_init, 10.0, 8)
    for g, r in zip(got[:3], ref[:3]):
        assert rel(g, r) < 5e-6
    assert abs(got[3] - ref[3]) < 1e-6 * abs(ref[3])
    want = oracle.cpd_estep_fgt(y, x, 0.3, s2, s2_init, 10.0, 8.0)
    for g, w in zip(got[:3], want[:3]):
        assert rel(g, w) < ESTEP_TOL
    assert abs(got[3] - want[3]) < 2e-5 * abs(want[3])         # (the oracle adds 120 000 logs one by one in fp32, as cpu-slam does: ~1e-5 of the sum is its own rounding)


@pytest.mark.parametrize("name", ["init", "s006"])
def test_bunny_fgt_estep_golden(ctx, golden, bunny, name):
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    e = golden.npz("bunny_fgt_estep.npz")
    c = g["esteps"][name]
    st = g["stride"]
    p1, pt1, px, L = ctx.cpd_estep_fgt(before, after, g["weight"], c["sigma2"], g["sigma2_init"])
    assert rel(p1[::st], e[name + "_p1"]) < ESTEP_TOL
    assert rel(pt1[::st], e[name + "_pt1"]) < ESTEP_TOL
    assert rel(px[::st], e[name + "_px"]) < ESTEP_TOL
    # cpu-slam sums the 14 904 logs sequentially in fp32 (cpdutils.cpp:69-71): its own rounding is ~1e-5 of the sum; ours is fp64
    assert abs(L - c["L"]) < 1e-4 * abs(c["L"])
    assert abs(float(p1.astype(np.float64).sum()) - c["p1_sum"]) < 1e-5 * c["p1_sum"]


@pytest.mark.parametrize("seed,m,n,s2", [(0, 400, 500, 2.0), (1, 700, 300, 0.05), (2, 3000, 2000, 0.02)])
def test_truncated_estep_matches_oracle(ctx, oracle, seed, m, n, s2):
    y, x = pair(seed, m, n)
    c = oracle.cpd_constant(4.0, 0.3, m, n)
    want = oracle.cpd_estep_truncated(y, x, c, s2, 1e-3)
    got = ctx.cpd_estep_truncated(y, x, c, s2, 1e-3)
    # an affinity within an ulp of the cut (exp(index) vs 1e-3) may fall on the other side: each costs <= 1e-3/den of one entry
    for g, w in zip(got[:3], want[:3]):
        assert rel(g, w) < 2e-4
    assert abs(got[3] - want[3]) < 1e-5 * abs(want[3]) + 1e-3


def test_bunny_truncated_estep_golden(ctx, golden, bunny):
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    e = golden.npz("bunny_fgt_estep.npz")
    t = g["truncated"]
    st = g["stride"]
    p1, pt1, px, L = ctx.cpd_estep_truncated(before, after, g["constant"], t["sigma2"], t["truncate"])
    assert rel(p1[::st], e["trunc_p1"]) < 2e-4 and rel(pt1[::st], e["trunc_pt1"]) < 2e-4 and rel(px[::st], e["trunc_px"]) < 2e-4
    assert abs(L - t["L"]) < 1e-4 * abs(t["L"])              # sequential fp32 log sum there, fp64 here


@pytest.fixture(scope="module")
def every_pair_ctx(capi):
    """A context whose truncated E-step is round 4's every-pair kernel (MISLAM_CPD_TRUNC_CULL=0; switches are read at context creation)."""
    import os
    os.environ["MISLAM_CPD_TRUNC_CULL"] = "0"
    try:
        c = capi.Context(0)
    finally:
        del os.environ["MISLAM_CPD_TRUNC_CULL"]
    yield c
    c.close()


@pytest.mark.parametrize("seed,m,n,s2", [(3, 1, 1, 1.0), (4, 1, 700, 0.3), (5, 65, 63, 0.05), (6, 64, 64, 5.0), (7, 1025, 1023, 0.01),
                                         (8, 5000, 9000, 0.004), (9, 20000, 17000, 0.02), (10, 40000, 30000, 1e-4)])
def test_culled_truncated_estep_equals_the_every_pair_one(ctx, every_pair_ctx, oracle, seed, m, n, s2):
    # K7t (cpd_trunc.hip): tiles of 64 points along a space-filling curve, skipped when their boxes are farther apart than the truncation
    # radius.  A skipped pair contributes exactly 0.0f in the reference (coherentpointdrift.cpp:193-196), so the culled sums hold the same
    # non-zero terms as the every-pair kernel's, in another order: equal to fp32 summation rounding.  Ragged sizes (one point, one lane short of
    # a tile, one past a super-tile), a radius that reaches everything (s2 = 5: nothing is skipped) and one that reaches almost nothing (1e-4)
    y, x = pair(seed, m, n)
    c = oracle.cpd_constant(4.0, 0.3, m, n)
    a = ctx.cpd_estep_truncated(y, x, c, s2, 1e-3)
    b = every_pair_ctx.cpd_estep_truncated(y, x, c, s2, 1e-3)
    for g, w in zip(a[:3], b[:3]):
        assert rel(g, w) < 2e-5, (m, n, s2, rel(g, w))
    assert abs(a[3] - b[3]) < 1e-6 * abs(b[3]) + 1e-4
    if m * n <= 2000000:
        want = oracle.cpd_estep_truncated(y, x, c, s2, 1e-3)
        for g, w in zip(a[:3], want[:3]):
            assert rel(g, w) < 2e-4


def test_culled_truncated_estep_with_nothing_in_reach(ctx, oracle):
    # two clouds farther apart than the truncation radius: every affinity is cut -- den = c, Pt1 = 0, P1 = PX = 0, exactly
    rng = np.random.default_rng(21)
    y = rng.uniform(-1, 1, (900, 3)).astype(np.float32)
    x = (rng.uniform(-1, 1, (1100, 3)) + 50.0).astype(np.float32)
    c = oracle.cpd_constant(4.0, 0.3, 900, 1100)
    p1, pt1, px, L = ctx.cpd_estep_truncated(y, x, c, 0.5, 1e-3)
    o1, ot1, ox, oL = oracle.cpd_estep_truncated(y, x, c, 0.5, 1e-3)
    assert not p1.any() and not px.any() and np.array_equal(pt1, ot1) and abs(L - oL) < 1e-5 * abs(oL)


def test_hybrid_run_is_the_same_registration_with_and_without_culling(ctx, every_pair_ctx, capi, golden, bunny):
    # the whole hybrid run (FGT E-steps, then truncated ones): same iteration count, s*R|t within E-step summation rounding of each other
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    p = capi.cpd_params(max_iterations=50, sigma2_init=g["sigma2_init"], approximation=capi.CPD_APPROX_HYBRID)
    a, b = ctx.cpd_register(before, after, p), every_pair_ctx.cpd_register(before, after, p)
    assert a[3] == b[3] == 23
    assert frob(a[0], a[1], b[0], b[1]) < 2e-5


def test_fixed_side_clustering_beside_the_moving_side_changes_no_bit(ctx, capi, golden, bunny, monkeypatch):
    # Round 5: an FGT E-step clusters the fixed cloud on the context's auxiliary stream while the main stream works on the moving cloud
    # (MISLAM_FGT_TWO_STREAMS=0: one after the other, as rounds 1-4).  Same kernels, same inputs, another schedule: hybrid and full runs bit for bit.
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    monkeypatch.setenv("MISLAM_FGT_TWO_STREAMS", "0")
    with capi.Context(0) as one:
        for approx, cap in ((capi.CPD_APPROX_HYBRID, 50), (capi.CPD_APPROX_FULL, 17)):
            p = capi.cpd_params(max_iterations=cap, sigma2_init=g["sigma2_init"], approximation=approx)
            a, b = ctx.cpd_register(before, after, p), one.cpd_register(before, after, p)
            assert a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[4] == b[4]
            # and again on the same contexts (buffers and clusterings carried over from the run before)
            a2 = ctx.cpd_register(before, after, p)
            assert a2[3] == a[3] and np.array_equal(a2[0], a[0]) and np.array_equal(a2[1], a[1])


def test_member_lists_made_inside_the_model_kernel_change_no_bit(ctx, capi, golden, bunny, monkeypatch):
    # Round 5: for clouds of at most 32 768 points every cell's workgroup of the model build lists its own members (cpd_fgt.hip, fgt_model_kernel<.., LISTS>)
    # instead of the stable counting sort's three launches per side (MISLAM_FGT_LISTS_IN_MODEL=0).  The same lists, hence the same sums: E-step
    # arrays, hybrid and full runs bit for bit -- on the bunny clouds, on ragged sizes (1 .. 4 waves' ranges cut short, fewer points than lanes of a
    # workgroup), with labels that leave cells empty (duplicated points) and at an order of truncation whose model build takes several workgroups per cell.
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    monkeypatch.setenv("MISLAM_FGT_LISTS_IN_MODEL", "0")
    with capi.Context(0) as sort:
        for approx, cap in ((capi.CPD_APPROX_HYBRID, 50), (capi.CPD_APPROX_FULL, 17)):
            p = capi.cpd_params(max_iterations=cap, sigma2_init=g["sigma2_init"], approximation=approx)
            a, b = ctx.cpd_register(before, after, p), sort.cpd_register(before, after, p)
            assert a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[4] == b[4]
        for seed, (m, n) in enumerate(((2, 3), (63, 65), (64, 64), (300, 511), (513, 700), (1025, 4097), (5000, 3000))):
            y, x = pair(seed, m, n)
            if seed % 2 == 1:
                y[m // 2:] = y[: m - m // 2]                     # duplicated points: the sweep leaves cells without members
            for order in (8, 12):
                ra, rb = ctx.cpd_estep_fgt(y, x, 0.3, 0.7, 4.0, 10.0, order), sort.cpd_estep_fgt(y, x, 0.3, 0.7, 4.0, 10.0, order)
                for u, v in zip(ra, rb):
                    assert np.array_equal(np.asarray(u), np.asarray(v), equal_nan=True), (m, n, order)
        # many cells (K = 450: still listed in the model kernel; K = 2 050: K x n label reads would cost more than the sort's O(n) -- the sort takes over)
        y, x = pair(11, 5000, 3000)
        for sigma2 in (0.01, 0.002):
            ra, rb = ctx.cpd_estep_fgt(y, x, 0.3, sigma2, 4.0, 10.0, 8), sort.cpd_estep_fgt(y, x, 0.3, sigma2, 4.0, 10.0, 8)
            for u, v in zip(ra, rb):
                assert np.array_equal(np.asarray(u), np.asarray(v), equal_nan=True), sigma2


def test_estep_primitives_reject_bad_arguments(ctx, capi):
    y, x = pair(0, 50, 60)
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_estep_fgt(y, x, 0.3, 1.0, 4.0, 10.0, 0)          # order of truncation
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_estep_fgt(y, x, 0.3, 1.0, 4.0, 10.0, 17)
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_estep_fgt(y[:1], x, 0.3, 1.0, 4.0, 10.0, 8)      # one point
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_estep_fgt(y, x, 0.3, 0.0, 4.0, 10.0, 8)
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_estep_truncated(y, x, 1.0, 1.0, 0.0)


# ---------------------------------------------------------------------------------------------------------------------
# full runs
# ---------------------------------------------------------------------------------------------------------------------
def test_bunny_hybrid_run_matches_cpu_slam(ctx, capi, golden, bunny):
    # cfg: cpd with the parser's default approximation type.  north_star bar: R|t within 1e-4 (Frobenius) of cpu-slam
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    r = g["runs"]["hybrid"]
    p = capi.cpd_params(max_iterations=r["max_iterations"], sigma2_init=g["sigma2_init"], approximation=capi.CPD_APPROX_HYBRID)
    sR, t, sc, it, err = ctx.cpd_register(before, after, p)
    assert it == r["iterations"] == 23
    check_measured("bunny_hybrid_vs_cpu_slam", frob(sR, t, np.array(r["R"]), np.array(r["t"])), 1e-4)
    assert err < 1e-3


def test_resumed_clustering_changes_nothing(ctx, capi, golden, bunny, monkeypatch):
    monkeypatch.setenv("MISLAM_FGT_RESUME", "0")       # read at context creation
    with capi.Context(0) as fresh:
        _resumed_clustering_changes_nothing(ctx, fresh, capi, golden, bunny)


def _resumed_clustering_changes_nothing(ctx, fresh, capi, golden, bunny):
    # the fixed cloud's K-centre sweep is resumed from one E-step to the next (K only grows as sigma^2 shrinks); switching that
    # off re-clusters from scratch every time and must give the same bits
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    p = capi.cpd_params(max_iterations=50, sigma2_init=g["sigma2_init"], approximation=capi.CPD_APPROX_HYBRID)
    a = ctx.cpd_register(before, after, p)
    b = fresh.cpd_register(before, after, p)
    assert a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[4] == b[4]
    # full mode: sigma^2 (hence K) also moves down again once the clamp kicks in -- the sweep restarts there
    p = capi.cpd_params(max_iterations=24, sigma2_init=g["sigma2_init"], approximation=capi.CPD_APPROX_FULL)
    a = ctx.cpd_register(before, after, p)
    b = fresh.cpd_register(before, after, p)
    assert a[3] == b[3] == 24 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[4] == b[4]


def test_few_moving_points_against_many_fixed_ones_resume_changes_no_bit(ctx, capi, monkeypatch):
    # 24 moving points against 32 768 fixed ones is K <= 24 cells of 1 000+ members on the fixed side -- big enough for the model build's
    # split over workgroups (Z > 1), small enough for the cells to list their own members (one workgroup per cell).  Which of the two ran used to depend
    # on whether THAT E-step re-clustered the fixed cloud (two orders of the same sums: bits that moved between iterations and between RESUME on / off);
    # now it follows from n, K and pd alone.
    rng = np.random.default_rng(12)
    a = (rng.normal(size=(32768, 3)) * np.array([1.5, 1.0, 0.6])).astype(np.float32)
    b = (a[rng.permutation(32768)[:24]] + rng.normal(scale=0.02, size=(24, 3)) + np.array([0.1, -0.05, 0.08])).astype(np.float32)
    monkeypatch.setenv("MISLAM_FGT_RESUME", "0")       # read at context creation
    with capi.Context(0) as fresh:
        for approx in (capi.CPD_APPROX_FULL, capi.CPD_APPROX_HYBRID):
            p = capi.cpd_params(max_iterations=6, tolerance=0.0, approximation=approx)
            r1, r2 = ctx.cpd_register(b, a, p), fresh.cpd_register(b, a, p)
            assert r1[3] == r2[3] and np.array_equal(r1[0], r2[0], equal_nan=True) and np.array_equal(r1[1], r2[1], equal_nan=True), approx
            r3 = ctx.cpd_register(b, a, p)                      # ... and a second run on the resumed context
            assert np.array_equal(r1[0], r3[0], equal_nan=True) and np.array_equal(r1[1], r3[1], equal_nan=True)


def test_large_cloud_sweep_and_its_resume(ctx, capi, oracle, monkeypatch):
    # above 65 536 points the sweep runs one grid-wide launch per centre; the fixed cloud's sweep is still resumed as K grows
    rng = np.random.default_rng(4)
    b = (rng.normal(size=(90000, 3)) * np.array([2.0, 1.0, 0.5])).astype(np.float32)
    ang = 0.25
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    a = (b[rng.permutation(90000)[:80000]] @ Rz.T + np.array([0.3, -0.2, 0.1])).astype(np.float32)
    lab_o = oracle.fgt_kcenter(a, 60)[1]
    assert np.array_equal(ctx.fgt_kcenter(a, 60)[1], lab_o)
    p = capi.cpd_params(max_iterations=8, tolerance=0.0, approximation=capi.CPD_APPROX_HYBRID)
    r1 = ctx.cpd_register(b, a, p)
    monkeypatch.setenv("MISLAM_FGT_RESUME", "0")       # read at context creation
    with capi.Context(0) as fresh:
        r2 = fresh.cpd_register(b, a, p)
    assert r1[3] == r2[3] == 8 and np.array_equal(r1[0], r2[0]) and np.array_equal(r1[1], r2[1]) and r1[4] == r2[4]
    assert np.abs(r1[0] / r1[2] - Rz).max() < 0.05             # sR / s: eight iterations in, the rotation is already there


@pytest.mark.parametrize("cap", [5, 17])
def test_bunny_full_mode_capped_matches_cpu_slam(ctx, capi, golden, bunny, cap):
    before, after = bunny
    g = golden.json("bunny_fgt.json")
    r = g["runs"]["full_cap%d" % cap]
    p = capi.cpd_params(max_iterations=cap, sigma2_init=g["sigma2_init"], approximation=capi.CPD_APPROX_FULL)
    sR, t, sc, it, err = ctx.cpd_register(before, after, p)
    assert it == cap
    assert frob(sR, t, np.array(r["R"]), np.array(r["t"])) < 1e-4
    assert abs(err - r["error"]) < 5e-4 * r["error"]


@pytest.mark.parametrize("approx", [1, 2])
def test_small_runs_match_oracle(ctx, capi, oracle, approx):
    rng = np.random.default_rng(11)
    b = (rng.normal(size=(2000, 3)) * 2).astype(np.float32)
    ang = 0.3
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    a = (b[rng.permutation(2000)[:1500]] @ Rz.T + np.array([0.4, -0.2, 0.1])).astype(np.float32)
    cap = 15
    Ro, to, ito, eo = oracle.cpd_approx(b, a, approx, max_iterations=cap)
    p = capi.cpd_params(max_iterations=cap, sigma2_init=oracle.cpd_sigma_squared(b, a), approximation=approx)
    sR, t, sc, it, err = ctx.cpd_register(b, a, p)
    assert it == ito
    assert frob(sR, t, Ro, to) < 1e-4
    assert abs(err - eo) < 2e-3 * max(eo, 1e-3)


def test_full_mode_clamps_sigma(ctx, capi, oracle):
    # approximation "full": sigma^2 never enters an E-step below 0.05 (coherentpointdrift.cpp:154-155).  On nearly aligned clouds the
    # first M-step already lands below that, so every later E-step runs at the clamp; the run must retrace the restatement's.
    b = cloud(2, 800, spread=0.15)                           # sigma^2_init = 0.046: clamped from the first E-step on
    a = (b + np.array([0.05, 0.0, -0.02])).astype(np.float32)
    cap = 6
    Ro, to, ito, eo, trace = oracle.cpd_approx(b, a, oracle.APPROX_FULL, max_iterations=cap, tolerance=1e-9, trace_cap=cap)
    p = capi.cpd_params(max_iterations=cap, tolerance=1e-9, sigma2_init=oracle.cpd_sigma_squared(b, a), approximation=capi.CPD_APPROX_FULL)
    sR, t, sc, it, err = ctx.cpd_register(b, a, p)
    assert it == ito == cap
    assert (trace[:, 0] < 0.05).all()                        # every M-step lands below the clamp
    assert frob(sR, t, Ro, to) < 1e-4
    assert abs(err - eo) < 1e-3 * eo


def test_register_rejects_bad_fgt_parameters(ctx, capi):
    b = cloud(0, 100)
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_register(b, b, capi.cpd_params(max_iterations=3, approximation=3))
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_register(b, b, capi.cpd_params(max_iterations=3, approximation=capi.CPD_APPROX_HYBRID, fgt_order_of_truncation=0))
    with pytest.raises(capi.MiSlamError):
        ctx.cpd_register(b[:1], b, capi.cpd_params(max_iterations=3, approximation=capi.CPD_APPROX_FULL))
