"""GPU suite: the ICP path through the C ABI.

Parity bars: correspondence indices, kept-pair sets and iteration counts bit-exact; R|t within 1e-4 Frobenius of the
reference's cpu-slam result (BASELINE.json north_star) -- in practice ~1e-5, the fp32-vs-fp64 summation difference.
"""
import numpy as np
import pytest

from conftest import check_measured, frob, synth_cloud

pytestmark = pytest.mark.gpu


def test_kabsch_matches_oracle_and_golden(ctx, capi, oracle, golden, bunny):
    before, after = bunny
    g = golden.npz("bunny_icp_iter0.npz")
    idx, d2 = ctx.nn_search(before, after)
    keep = (d2 < np.float32(400.0)).astype(np.uint8)
    R, t, used = ctx.kabsch(before, after, idx, keep)
    assert used == len(g["idx_before"])
    assert np.abs(R - g["R0"]).max() < 5e-6 and np.abs(t - g["t0"]).max() < 5e-6      # the reference's own first solve
    Ro, to = oracle.least_squares_svd(before[keep > 0], after[idx[keep > 0]])
    assert np.abs(R - Ro).max() < 5e-6 and np.abs(t - to).max() < 5e-6


@pytest.mark.parametrize("seed", range(4))
def test_kabsch_random_pairs(ctx, capi, oracle, seed):
    rng = np.random.default_rng(seed)
    n, m = int(rng.integers(3, 3000)), int(rng.integers(3, 3000))
    src = rng.normal(size=(n, 3)).astype(np.float32) * 3
    tgt = rng.normal(size=(m, 3)).astype(np.float32) * 3
    idx = rng.integers(0, m, n).astype(np.int32)
    keep = (rng.uniform(size=n) < 0.7).astype(np.uint8)
    keep[:3] = 1
    R, t, used = ctx.kabsch(src, tgt, idx, keep)
    Ro, to = oracle.least_squares_svd(src[keep > 0], tgt[idx[keep > 0]])
    assert used == int(keep.sum())
    assert np.abs(R - Ro).max() < 2e-5 and np.abs(t - to).max() < 1e-4
    assert abs(np.linalg.det(R.astype(np.float64)) - 1) < 1e-4


def test_kabsch_reflection_case(ctx, capi, oracle):
    rng = np.random.default_rng(11)
    b = rng.normal(size=(50, 3)).astype(np.float32)
    b[:, 2] = 0
    a = b.copy(); a[:, 0] = -a[:, 0]
    idx = np.arange(50, dtype=np.int32)
    R, t, _ = ctx.kabsch(b, a, idx)
    Ro, to = oracle.least_squares_svd(b, a)
    assert np.abs(R - Ro).max() < 1e-5 and np.linalg.det(R.astype(np.float64)) > 0.99


def test_transform_bit_exact_and_mse(ctx, capi, oracle):
    rng = np.random.default_rng(5)
    src = rng.uniform(-5, 5, (5003, 3)).astype(np.float32)
    tgt = rng.uniform(-5, 5, (4001, 3)).astype(np.float32)
    R = np.array([[0.36, 0.47, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32)
    t = np.array([1.0, -2.0, 0.5], np.float32)
    idx = rng.integers(0, 4001, 5003).astype(np.int32)
    keep = (rng.uniform(size=5003) < 0.5).astype(np.uint8)
    out, mse = ctx.transform_mse(src, R, t, tgt, idx, keep, divide_by_pairs=True)
    ref_cloud = oracle.transform_cloud(src, R, t)
    assert np.array_equal(out, ref_cloud)                                   # same operation order, no contraction
    ib = np.nonzero(keep)[0].astype(np.int32)
    ref_mse = oracle.mse_indexed(ref_cloud, tgt, ib, idx[ib])
    assert abs(mse - ref_mse) < 2e-6 * ref_mse                              # fp64 sum here, sequential fp32 there
    out2, mse2 = ctx.transform_mse(src, R, t, tgt, idx, None, divide_by_pairs=False)
    d = tgt[idx].astype(np.float64) - ref_cloud
    assert abs(mse2 - (d ** 2).sum() / 4001) < 1e-5 * mse2                  # cuda-slam rule: / after.size()


def test_bunny_icp_matches_cpu_slam(ctx, capi, oracle, golden, bunny):
    # cfg 1: config/default.json.  39 iterations, R|t within 1e-4 of the reference's cpu-slam run.
    before, after = bunny
    g = golden.json("bunny_icp.json")
    p = capi.icp_params(eps=1e-3, max_iterations=50, max_distance_squared=400.0)
    R, t, it, err = ctx.icp_register(before, after, p)
    assert it == g["iterations"] == 39
    d_ref = frob(R, t, g["R"], g["t"])
    check_measured("bunny_icp_vs_cpu_slam", d_ref, 1e-4, factor=1.5)
    assert abs(err - g["error"]) < 1e-6
    # against the oracle restatement run here, tighter
    Ro, to, ito, eo = oracle.icp(before, after, 1e-3, 400.0, 50)
    d_or = frob(R, t, Ro, to)
    print("bunny ICP |d(R|t)|_F vs cpu-slam = %.3e, vs oracle = %.3e" % (d_ref, d_or))
    assert ito == it
    check_measured("bunny_icp_vs_oracle", d_or, 2e-5, factor=1.5)


@pytest.mark.parametrize("k", [1, 2, 3, 5, 10, 20])
def test_bunny_icp_trajectory_matches_cpu_slam(ctx, capi, golden, bunny, k):
    before, after = bunny
    c = golden.json("bunny_icp.json")["capped"][str(k)]
    p = capi.icp_params(eps=1e-3, max_iterations=k, max_distance_squared=400.0)
    R, t, it, err = ctx.icp_register(before, after, p)
    assert it == c["iterations"]
    assert frob(R, t, c["R"], c["t"]) < 1e-4
    # (cpu-slam's error is ONE sequential fp32 sum over its 14 904 residuals: 1e-5 relative is that sum's own rounding level, and the
    # residuals themselves move with the ~1e-5 difference in R|t)
    assert abs(err - c["error"]) < 2e-5 * max(1.0, c["error"])


@pytest.mark.parametrize("sync_every", [1, 3, 8])
def test_result_independent_of_host_check_interval(ctx, capi, bunny, sync_every):
    before, after = bunny
    base = ctx.icp_register(before, after, capi.icp_params(max_iterations=50, max_distance_squared=400.0, sync_every=1))
    other = ctx.icp_register(before, after, capi.icp_params(max_iterations=50, max_distance_squared=400.0, sync_every=sync_every))
    assert base[2] == other[2] and np.array_equal(base[0], other[0]) and np.array_equal(base[1], other[1]) and base[3] == other[3]


@pytest.mark.parametrize("n", [3, 64, 65, 2000, 14904, 100000, 131072])
def test_one_launch_reduce_and_solve_is_the_two_launch_one(ctx, capi, monkeypatch, n):
    # up to 131 072 moving points (2 048 rows) the rows reduce and the deferred solve are ONE launch of one workgroup (round 4) that adds
    # the rows up in the two-launch form's order: same bits (MISLAM_ICP_FUSED_SOLVE=0, read at context creation, keeps two launches)
    before, after = synth_cloud(n, seed=n)[:2]
    monkeypatch.setenv("MISLAM_ICP_FUSED_SOLVE", "0")
    with capi.Context(0) as two:
        for nn in ((capi.NN_GRID, capi.NN_BRUTEFORCE) if n <= 20000 else (capi.NN_GRID,)):
            p = capi.icp_params(eps=0.0, max_iterations=7, nn_mode=nn)
            a, b = ctx.icp_register(before, after, p), two.icp_register(before, after, p)
            assert a[2] == b[2] == 7 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3], (n, nn)


@pytest.mark.parametrize("n", [100, 5000, 14904, 100000, 300000])
def test_walks_split_over_two_waves_change_nothing(capi, monkeypatch, n):
    # fused iterations of small clouds (up to 450 000 moving points) run a helper wave per workgroup that takes the walks -- of the lanes
    # beyond the grid's reach while the first wave scans for the others, or of half the lanes of a chunk that walks at once -- and hands its
    # answers to the first wave (round 4).  MISLAM_GRID_SPLIT_WALKS=0 / 1 (read at context creation) forces one / two waves at any size:
    # ten iterations from a start far enough out that most chunks walk, then converging -- the same bits
    before, after = synth_cloud(n, seed=n + 1)[:2]
    out = []
    for split in ("0", "1", "2"):                                      # 2: helpers only beside a scan (the default between 200 000 and 450 000 points)
        monkeypatch.setenv("MISLAM_GRID_SPLIT_WALKS", split)
        with capi.Context(0) as c2:
            out.append((c2.icp_register(before, after, capi.icp_params(eps=0.0, max_iterations=10)),
                        c2.icp_register(before, after, capi.icp_params(cuda_slam=True, eps=0.0, max_iterations=3, dist_mode=1))))
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3]


@pytest.mark.parametrize("cuda_rules", [False, True])
def test_pipelined_host_checks_change_nothing(ctx, capi, bunny, monkeypatch, cuda_rules):
    # Round 4: with batches of more than one iteration an intermediate host check only peeks at the state (copied behind the batch, one
    # iteration of the next batch already behind the copy) instead of settling the pending iteration and draining the stream
    # (MISLAM_ICP_PIPELINE=0, read at context creation: the old way).  Converged and capped runs, every batch size, and budgets handed to
    # mi_icp_run piece by piece must give the same iterations, transform and error bit for bit -- and never run more than they were asked to.
    before, after = bunny
    monkeypatch.setenv("MISLAM_ICP_PIPELINE", "0")
    with capi.Context(0) as plain:
        for cap in (50, 7, 1):
            want = plain.icp_register(before, after, capi.icp_params(cuda_slam=cuda_rules, max_iterations=cap, max_distance_squared=400.0, sync_every=1))
            for sync in (0, 2, 3, 5, 16, 64):
                for c2 in (ctx, plain):
                    got = c2.icp_register(before, after, capi.icp_params(cuda_slam=cuda_rules, max_iterations=cap, max_distance_squared=400.0, sync_every=sync))
                    assert got[2] == want[2] and np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[3] == want[3], (cap, sync)
        p = capi.icp_params(cuda_slam=cuda_rules, eps=0.0, max_iterations=-1, max_distance_squared=400.0, sync_every=4)
        for c2 in (ctx, plain):
            c2.icp_load(before, after, p)
        ran = 0
        for budget in (1, 2, 3, 4, 5, 9, 11):
            a, b = ctx.icp_run(budget), plain.icp_run(budget)
            ran += budget
            ra, rb = ctx.icp_result(), plain.icp_result()
            assert a == b == budget and ra[2] == rb[2] == ran and np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and ra[3] == rb[3], budget
        ctx.icp_reset()


@pytest.mark.parametrize("cuda_rules", [False, True])
def test_search_strategy_does_not_change_the_registration(ctx, capi, bunny, cuda_rules):
    # every pair, the box hierarchy and the cell grid return the same keys, and every path -- the grid's fused iteration
    # included -- adds an iteration's sums in the same per-64-point rows: the whole run is bitwise identical
    before, after = bunny
    runs = []
    for nn_mode in (capi.NN_BRUTEFORCE, capi.NN_TREE, capi.NN_GRID):
        p = capi.icp_params(cuda_slam=cuda_rules, max_iterations=25, max_distance_squared=400.0, nn_mode=nn_mode)
        runs.append(ctx.icp_register(before, after, p))
    a = runs[0]
    for b in runs[1:]:
        assert a[2] == b[2] and a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_synth2k_matches_cpu_slam(ctx, capi, golden):
    z = golden.npz("synth2k_clouds.npz")
    g = golden.json("synth2k_icp.json")
    R, t, it, err = ctx.icp_register(z["before"], z["after"], capi.icp_params(max_iterations=60))
    assert it == g["iterations"]
    assert frob(R, t, g["R"], g["t"]) < 1e-4


@pytest.mark.parametrize("seed", [143, 158, 162, 241, 247, 353, 365])
def test_planar_and_duplicated_clouds_stay_finite_and_match_the_oracle(ctx, capi, oracle, seed):
    # Round 4 (found by tools/nn_soak.py): a PLANAR moving cloud gives a cross-covariance with an exactly zero column; a 2 x 2 block of
    # the Jacobi sweep is then symmetrised to rounding residue, the rotation angle's quotient overflows -- IEEE arithmetic turns that into
    # the identity rotation, the fast reciprocal / root forms of K3 turned it into inf - inf.  Seven of 400 random small problems (these).
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from nn_soak import cloud
    rng = np.random.default_rng(seed)
    n = int(10 ** rng.uniform(1, 3.5)); m = int(10 ** rng.uniform(1, 3.5))
    kt, ks = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    tgt = cloud(rng, m, kt).astype(np.float32); src = cloud(rng, n, ks).astype(np.float32)
    Ro, to, ito, eo = oracle.icp(src, tgt, eps=0.0, max_iterations=3)[:4]
    for nn in (capi.NN_BRUTEFORCE, capi.NN_TREE, capi.NN_GRID):
        R, t, it, err = ctx.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=3, nn_mode=nn))[:4]
        assert np.isfinite(R).all() and np.isfinite(t).all() and np.isfinite(err)
        assert it == ito == 3
        # (2e-4: three iterations on a rank-deficient problem; what each seed measures is recorded and held to 2x: tests/golden/measured_bounds.json)
        check_measured("planar_seed%d_nn%d_vs_oracle" % (seed, nn), frob(R, t, Ro, to) / max(1.0, float(np.abs(to).max())), 2e-4, floor=2e-6)


@pytest.fixture(scope="module")
def ieee_ctx(capi):
    """A context whose 3 x 3 SVDs run in IEEE divisions and roots (MISLAM_SVD_IEEE=1; switches are read at context creation)."""
    import os
    os.environ["MISLAM_SVD_IEEE"] = "1"
    try:
        c = capi.Context(0)
    finally:
        del os.environ["MISLAM_SVD_IEEE"]
    yield c
    c.close()


def soak_problem(seed, case):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from reg_soak import problems
    for k, degenerate, n, m, ks, kt, src, tgt in problems(case + 1, seed):
        if k == case:
            return src, tgt
    raise AssertionError("no such case")


@pytest.mark.parametrize("seed,case,bar", [(1, 174, 2e-5), (1, 333, None), (3, 337, None)])
def test_soak_cases_root_caused_in_round_5(ctx, ieee_ctx, capi, oracle, seed, case, bar):
    # Round 4's registration soak filed its worst "well-posed" differences from the oracle (2.3e-3 ICP, 3.7e-3 CPD after 3 / 5 iterations) under
    # "clustered against clustered".  Round 5 took them apart (tools/soak_rootcause.py, profiles/r05_soak_rootcause.log):
    #  * seed 1 case 174 (two tight clusters against a plane; singular values of the cross-covariance 1990 / 0.106 / 0): a RANK-DEFICIENT H.  The
    #    null direction's sign -- rotation or its mirror image -- is decided by the last bits of the Jacobi sweep; Eigen's arithmetic (the oracle,
    #    reproducible to 3e-6 with the points reordered) takes one branch, K3's refined hardware forms took the other: |d(R|t)|_F = 1.81.  K3 now
    #    hands a decomposition whose smallest singular value is below 1e-3 of the largest to the IEEE forms (svd3.hpp): 4e-6.
    #  * seed 1 case 333, seed 3 case 337: differences of 5e-2 / 1.5 that the ORACLE shows against itself when the moving cloud is merely
    #    reordered (cpu-slam's sequential fp32 centroid sums round differently): the problem's own conditioning, no kernel's error.  Asserted
    #    as such: the device is no farther from the oracle than 2 x the oracle from itself -- the maximum over twelve seeded reorderings, PINNED in
    #    tests/golden/soak_spread.json (oracle/make_golden_soak_spread.py; round 5 drew the twelve at run time: a noisy bar).
    src, tgt = soak_problem(seed, case)
    Ro, to, ito, eo = oracle.icp(src, tgt, eps=0.0, max_iterations=3)[:4]
    scale = max(1.0, float(np.abs(to).max()))
    R, t, it, err = ctx.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=3))[:4]
    d = frob(R, t, Ro, to) / scale
    Ri, ti = ieee_ctx.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=3))[:2]
    print("soak seed %d case %d: |d(R|t)|_F / scale vs oracle: default %.3e, IEEE K3 %.3e" % (seed, case, d, frob(Ri, ti, Ro, to) / scale))
    assert it == ito == 3 and np.isfinite(R).all()
    if bar is not None:
        assert d < bar
    else:
        import json, os
        pinned = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "soak_spread.json")))["cases"]["%d/%d" % (seed, case)]
        assert abs(pinned["scale"] - scale) < 1e-6 * scale          # (the same problem as the fixture's)
        own = pinned["spread_max"]
        assert d <= 2.0 * own + 1e-5, (d, own)


def test_fast_and_ieee_k3_agree_on_the_fixtures(ctx, ieee_ctx, capi, golden, bunny):
    # the one change that moves results away from the reference's arithmetic is the FAST K3 (rcp / rsq refined by fmas instead of IEEE
    # divisions and roots).  A/B on one device, same inputs: bunny ICP (39 iterations composed) and bunny CPD (27 EM iterations) -- same iteration
    # counts, and each form as close to cpu-slam as the other (measured round 5: ICP 1.4e-5 fast / 1.2e-5 IEEE; the bar of the IEEE form, 1e-5 + what
    # the forms differ by, is kept for both)
    before, after = bunny
    g = golden.json("bunny_icp.json")
    p = g["params"]
    prm = capi.icp_params(eps=p["eps"], max_iterations=p["max_iterations"], max_distance_squared=p["max_distance_squared"])
    a, b = ctx.icp_register(before, after, prm), ieee_ctx.icp_register(before, after, prm)
    assert a[2] == b[2] == g["iterations"]
    d_ab = frob(a[0], a[1], b[0], b[1])
    d_fast, d_ieee = frob(a[0], a[1], g["R"], g["t"]), frob(b[0], b[1], g["R"], g["t"])
    print("bunny ICP vs cpu-slam: fast K3 %.3e, IEEE K3 %.3e, fast vs IEEE %.3e" % (d_fast, d_ieee, d_ab))
    assert d_ab < 1e-5 and d_ieee < 1.5e-5 and d_fast < 2e-5
    gc = golden.json("bunny_cpd.json")
    pc = capi.cpd_params(max_iterations=50, const_scale=0, sigma2_init=gc["sigma2_init"])
    a, b = ctx.cpd_register(before, after, pc), ieee_ctx.cpd_register(before, after, pc)
    f = gc["final_scale_free"]
    assert a[3] == b[3] == f["iterations"]
    print("bunny CPD vs cpu-slam: fast K3 %.3e, IEEE K3 %.3e, fast vs IEEE %.3e" % (frob(a[0], a[1], f["sR"], f["t"]), frob(b[0], b[1], f["sR"], f["t"]), frob(a[0], a[1], b[0], b[1])))
    assert frob(a[0], a[1], b[0], b[1]) < 2e-5


def test_cuda_slam_driver_rules(ctx, capi, oracle, golden):
    # exact composition, no filter, error / |after|, abort + rollback, FMA distance (icpcuda.cu:8-58) vs the oracle in
    # the same modes ("parity unpinned" against a real CUDA run: none can be made here)
    z = golden.npz("synth2k_clouds.npz")
    p = capi.icp_params(cuda_slam=True, max_iterations=60)
    R, t, it, err = ctx.icp_register(z["before"], z["after"], p)
    Ro, to, ito, eo = oracle.icp(z["before"], z["after"], 1e-3, 1000.0, 60, dist_mode=oracle.DIST_FMA,
                                 compose_mode=oracle.COMPOSE_EXACT, abort_on_increase=True, filter_pairs=False)
    assert it == ito
    assert frob(R, t, Ro, to) < 2e-5
    assert frob(R, t, z["R_true"], z["t_true"]) < 5e-2


def test_edge_cases(ctx, capi, bunny):
    before, after = bunny
    # max_iterations = 0: the loop never runs (identity, error 1e5)
    R, t, it, err = ctx.icp_register(before[:100], after[:100], capi.icp_params(max_iterations=0))
    assert it == 0 and err == pytest.approx(1e5) and np.array_equal(R, np.eye(3)) and np.all(t == 0)
    # nothing survives the distance filter: "if (correspondingPoints.size() == 0) break"
    ctx.icp_load(before[:100], after[:100] + 1000.0, capi.icp_params(max_iterations=5, max_distance_squared=1e-6))
    ctx.icp_run(-1)
    R, t, it, err, why = ctx.icp_result()
    assert why == capi.STOP_NO_PAIRS and it == 0 and np.array_equal(R, np.eye(3))
    # ragged sizes, single point target
    R, t, it, err = ctx.icp_register(before[:257], after[:1], capi.icp_params(max_iterations=3))
    assert np.all(np.isfinite(R)) and np.all(np.isfinite(t))
    with pytest.raises(capi.MiSlamError):
        ctx.icp_register(before[:0], after, capi.icp_params())
    with pytest.raises(capi.MiSlamError):
        ctx.icp_register(before, after, capi.icp_params(dist_mode=9))


def test_resident_stepping_equals_one_shot(ctx, capi, bunny):
    before, after = bunny
    p = capi.icp_params(max_iterations=50, max_distance_squared=400.0)
    one = ctx.icp_register(before, after, p)
    ctx.icp_load(before, after, p)
    total = 0
    while True:
        total += ctx.icp_run(7)
        R, t, it, err, why = ctx.icp_result()
        if why != capi.STOP_RUNNING:
            break
    assert why == capi.STOP_CONVERGED and it == one[2] and np.array_equal(R, one[0]) and np.array_equal(t, one[1])
    ctx.icp_reset()
    assert ctx.icp_result()[2] == 0


def test_cfg2_size_properties(ctx, capi, oracle):
    # cfg 2 (N = 1e5 synthetic): recovers the known rigid motion; the first correspondences equal the oracle's on sampled
    # rows; iterating is monotone in error until convergence.
    before, after, Rt, tt = synth_cloud(100000)
    p = capi.icp_params(max_iterations=40)
    ctx.icp_load(before, after, p)
    errs, dists = [], []
    while True:
        ctx.icp_run(1)
        R, t, it, err, why = ctx.icp_result()
        errs.append(err)
        dists.append(frob(R, t, Rt, tt))
        if why != capi.STOP_RUNNING:
            break
    # uniform random clouds converge slowly (point-to-point ICP): the run ends on the iteration cap, like the oracle would
    assert why in (capi.STOP_CONVERGED, capi.STOP_MAX_ITERATIONS) and len(errs) <= 40
    assert all(b <= a * (1 + 1e-5) for a, b in zip(errs, errs[1:]))          # error never rises
    assert errs[-1] < 0.05 * errs[0] and dists[-1] < 0.25 * dists[0]          # and the known motion is being recovered
    # first iteration on a 20 000-point slice against the oracle and against an fp64 Kabsch solve of the same pairs.
    # cpu-slam's centroid is ONE sequential fp32 running sum (common.cpp:281-284): at 2e4 points of magnitude ~10 it is
    # already 1.3e-4 off the true mean (measured; it grows with N), and t = c_after - R c_before inherits that.  The HIP
    # path sums in fp64, so it must sit within rounding of the fp64 solve and within that documented noise of the oracle.
    nb, na = before[:20000], after[:20000]
    ctx.icp_load(nb, na, capi.icp_params(max_iterations=1))
    ctx.icp_run(-1)
    R1, t1, it1, e1, _ = ctx.icp_result()
    Ro, to, ito, eo = oracle.icp(nb, na, 1e-3, 1000.0, 1)
    assert it1 == ito == 1 and np.abs(R1 - Ro).max() < 2e-6
    check_measured("cfg2_slice_first_iteration_vs_oracle", frob(R1, t1, Ro, to), 5e-4, factor=1.5)     # (cpu-slam's fp32 centroid: see above)
    idx, _ = ctx.nn_search(nb, na)
    B, A = nb.astype(np.float64), na[idx].astype(np.float64)
    cb, ca = B.mean(0), A.mean(0)
    U, S, Vt = np.linalg.svd((A - ca).T @ (B - cb))
    R64 = U @ np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    assert frob(R1, t1, R64, ca - R64 @ cb) < 5e-6


def test_ten_million_icp_iterations(ctx, capi, oracle):
    # cfg 5's size (N = M = 1e7) through the whole ICP loop, not just the search: three iterations on the default path (fused
    # cell-grid search): the error falls, the registration after ONE iteration equals an fp64 Kabsch solve of its own pairs,
    # sampled rows of those pairs are the oracle's nearest neighbours, and the loop is bitwise reproducible
    before, after, Rt, tt = synth_cloud(10000000)
    ctx.icp_load(before, after, capi.icp_params(max_iterations=3))
    errs = []
    for _ in range(3):
        ctx.icp_run(1)
        errs.append(ctx.icp_result()[3])
    R3, t3, it3, e3, why = ctx.icp_result()
    assert it3 == 3 and errs[0] > errs[1] > errs[2] > 0
    R1, t1, it1, e1 = ctx.icp_register(before, after, capi.icp_params(max_iterations=1))
    idx, d2 = ctx.nn_search(before, after)                       # the pairs of iteration 0 (identity start)
    rows = np.random.default_rng(5).choice(len(before), 48, replace=False)
    ridx, rd2 = oracle.nn_search(before[rows], after)
    assert np.array_equal(idx[rows], ridx) and np.array_equal(d2[rows].view(np.uint32), rd2.view(np.uint32))
    B, A = before.astype(np.float64), after[idx].astype(np.float64)
    cb, ca = B.mean(0), A.mean(0)
    U, S, Vt = np.linalg.svd((A - ca).T @ (B - cb))
    R64 = U @ np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    assert frob(R1, t1, R64, ca - R64 @ cb) < 2e-5
    assert abs(e1 - errs[0]) == 0
    again = ctx.icp_register(before, after, capi.icp_params(max_iterations=3))
    assert np.array_equal(again[0], R3) and np.array_equal(again[1], t3) and again[3] == e3


def test_chunk_classes_do_not_change_the_registration(ctx, capi):
    # The fused search keeps a class per 64-point chunk from one iteration to the next (no lane walked / some did / most lanes lie
    # beyond the grid's reach -> the chunk skips the scan and walks): scheduling only.  A partially overlapping pair of clouds takes
    # every chunk through the classes as the registration converges; every iteration's error and the final transform must equal
    # the every-pair search's bit for bit.
    before, after, Rt, tt = synth_cloud(60000)
    after = (after + np.float32(1.5)).astype(np.float32)                 # a third of the moving cloud starts outside the fixed one
    trail = {}
    for nn_mode in (capi.NN_GRID, capi.NN_BRUTEFORCE):
        ctx.icp_load(before, after, capi.icp_params(eps=0.0, max_iterations=25, nn_mode=nn_mode))
        errs = []
        for _ in range(25):
            ctx.icp_run(1)
            errs.append(ctx.icp_result()[3])
        trail[nn_mode] = (errs, ctx.icp_result())
    (eg, rg), (eb, rb) = trail[capi.NN_GRID], trail[capi.NN_BRUTEFORCE]
    assert eg == eb and np.array_equal(rg[0], rb[0]) and np.array_equal(rg[1], rb[1]) and rg[2] == rb[2] == 25
    assert eg[-1] < 0.2 * eg[0]


def test_full_bench_size_iterations(ctx, capi):
    # BASELINE.json's headline configuration (N = M = 1e6, the bench workload): the first iterations through both searches are
    # bitwise the same registration, the error falls monotonically, and one iteration equals an fp64 Kabsch solve of its pairs
    before, after, Rt, tt = synth_cloud(1000000)
    runs = []
    for nn_mode in (capi.NN_GRID, capi.NN_TREE, capi.NN_BRUTEFORCE):
        ctx.icp_load(before, after, capi.icp_params(max_iterations=3, nn_mode=nn_mode))
        errs = []
        for _ in range(3):
            ctx.icp_run(1)
            errs.append(ctx.icp_result()[3])
        runs.append((ctx.icp_result(), errs))
    (ra, ea) = runs[0]
    for rb, eb in runs[1:]:
        assert ea == eb and np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])
    assert ea[0] > ea[1] > ea[2]
    R1, t1, it1, e1 = ctx.icp_register(before, after, capi.icp_params(max_iterations=1))
    idx, _ = ctx.nn_search(before, after)
    B, A = before.astype(np.float64), after[idx].astype(np.float64)
    cb, ca = B.mean(0), A.mean(0)
    U, S, Vt = np.linalg.svd((A - ca).T @ (B - cb))
    R64 = U @ np.diag([1.0, 1.0, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    assert it1 == 1 and frob(R1, t1, R64, ca - R64 @ cb) < 5e-6
    assert abs(e1 - float((np.linalg.norm(A - (B @ R1.astype(np.float64).T + t1), axis=1) ** 2).mean())) < 1e-5 * e1


@pytest.mark.parametrize("n,iters", [(20000, 1), (20000, 6), (40000, 3), (100000, 2)])   # the last one is cfg 2's size
def test_cpu_sequential_sums_retrace_cpu_slam_beyond_bunny_size(ctx, capi, oracle, n, iters):
    # MI_SUM_CPU_SEQUENTIAL: cpu-slam's sequential fp32 centroid / error sums reproduced bit for bit.  With them the HIP path
    # follows the oracle's trajectory at sizes where the exact-sum default is 1.3e-4+ away (see test_cfg2_size_properties):
    # R|t within 1e-5 per iteration, error within 2e-5 relative.
    before, after, _, _ = synth_cloud(100000)
    nb, na = before[:n], after[:n]
    Ro, to, ito, eo = oracle.icp(nb, na, 1e-3, 1000.0, iters)
    R, t, it, err = ctx.icp_register(nb, na, capi.icp_params(max_iterations=iters, sum_mode=capi.SUM_CPU_SEQUENTIAL))
    assert it == ito
    d = frob(R, t, Ro, to)
    print("n=%d iters=%d |d(R|t)|_F vs oracle with cpu-slam's sums = %.3e" % (n, iters, d))
    assert d < 1e-5 * iters + 5e-6           # measured: 4.8e-7, 2.0e-6, 6.7e-6, 1.3e-5 for the four cases
    assert abs(err - eo) <= 2e-5 * eo        # the residuals themselves move with the ~1e-6 difference in R|t
    Re, te, _, _ = ctx.icp_register(nb, na, capi.icp_params(max_iterations=iters))
    assert frob(Re, te, Ro, to) > d          # the default (exact sums) is farther from cpu-slam than its own arithmetic


def test_cpu_sequential_sums_on_bunny(ctx, capi, golden, bunny):
    before, after = bunny
    g = golden.json("bunny_icp.json")
    p = capi.icp_params(eps=1e-3, max_iterations=50, max_distance_squared=400.0, sum_mode=capi.SUM_CPU_SEQUENTIAL)
    R, t, it, err = ctx.icp_register(before, after, p)
    d = frob(R, t, g["R"], g["t"])
    print("bunny ICP with cpu-slam's sums: |d(R|t)|_F vs cpu-slam = %.3e" % d)
    assert it == g["iterations"] and d < 1e-5 and abs(err - g["error"]) < 1e-7


def test_world1_rccl_context_matches_plain_context(capi, bunny):
    # the multi-GPU code path (RCCL communicator, packed-min all-reduce, shard-local moments + sum all-reduce) with one rank
    before, after = bunny
    uid = capi.dist_unique_id()
    with capi.Context(0, 0, 1, uid) as dctx, capi.Context(0) as sctx:
        assert dctx.rank_world() == (0, 1)
        for nn_mode in (capi.NN_BRUTEFORCE, capi.NN_TREE, capi.NN_GRID):
            for shard_mode in (capi.SHARD_TARGET, capi.SHARD_SOURCE, capi.SHARD_AUTO):
                p = capi.icp_params(max_iterations=12, max_distance_squared=400.0, nn_mode=nn_mode, shard_mode=shard_mode)
                a = dctx.icp_register(before, after, p)
                b = sctx.icp_register(before, after, p)
                assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (nn_mode, shard_mode)
        i1, d1 = dctx.nn_search(before[:1000], after)
        i2, d2 = sctx.nn_search(before[:1000], after)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
        # The multi-GPU path settles an iteration's stop rule one all-reduce later (its error sums ride with the next iteration's
        # moments; a flush closes every host batch).  Same iteration counts, errors and stop reasons as the plain context: a run
        # to convergence, the GPU reference's rules (abort + rollback on an error increase), the iteration cap, any host batch
        # size, and a run enqueued in pieces.
        for kw in (dict(eps=1e-3, max_iterations=50, max_distance_squared=400.0), dict(cuda_slam=True, max_iterations=60),
                   dict(max_iterations=7), dict(eps=1e-3, max_iterations=50, max_distance_squared=400.0, sync_every=1),
                   dict(eps=1e-3, max_iterations=50, max_distance_squared=400.0, sync_every=3)):
            cuda = kw.pop("cuda_slam", False)
            a = dctx.icp_register(before, after, capi.icp_params(cuda_slam=cuda, **kw))
            b = sctx.icp_register(before, after, capi.icp_params(cuda_slam=cuda, **kw))
            assert a[2] == b[2] and a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), kw
        for c in (dctx, sctx):
            c.icp_load(before, after, capi.icp_params(eps=1e-3, max_iterations=50, max_distance_squared=400.0))
            assert c.icp_run(5) == 5
            mid = c.icp_result()
            assert mid[2] == 5 and mid[4] == capi.STOP_RUNNING
            c.icp_run(-1)
        ra, rb = dctx.icp_result(), sctx.icp_result()
        assert ra[2] == rb[2] == 39 and ra[3] == rb[3] and ra[4] == rb[4] == capi.STOP_CONVERGED
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])


def test_multiplication_known_answer(ctx, capi):
    # MultiplicationTest, source/cuda-slam/cudacommon.cu:319-343: ones(3 x 100) * ones(100 x 3) through the reference's GEMM wrapper must
    # be 100 everywhere (+- 1e-5).  Here the product alignedAfter * alignedBefore^T of LeastSquaresSVD (cudacommon.cu:196-201) is the
    # nine cross sums of the moment rows: 100 pairs of (1, 1, 1) against (1, 1, 1).
    ones = np.ones((100, 3), np.float32)
    mom = ctx.cross_moments(ones, ones, np.arange(100, dtype=np.int32))
    assert mom[0] == 100
    assert np.all(np.abs(mom[7:16] - 100.0) <= 1e-5) and np.all(np.abs(mom[1:7] - 100.0) <= 1e-5)
    # the same product on values that are not all alike, against numpy in fp64; and with dropped pairs
    rng = np.random.default_rng(5)
    src = rng.uniform(-5, 5, (1003, 3)).astype(np.float32)
    tgt = rng.uniform(-5, 5, (777, 3)).astype(np.float32)
    idx = rng.integers(0, 777, 1003).astype(np.int32)
    keep = (rng.uniform(0, 1, 1003) < 0.8).astype(np.uint8)
    for kp in (None, keep):
        mom = ctx.cross_moments(src, tgt, idx, kp)
        sel = np.ones(1003, bool) if kp is None else kp.astype(bool)
        b, a = src[sel].astype(np.float64), tgt[idx[sel]].astype(np.float64)
        want = np.concatenate([[sel.sum()], b.sum(0), a.sum(0), (a[:, :, None] * b[:, None, :]).sum(0).reshape(9)])
        assert np.allclose(mom, want, rtol=1e-12, atol=1e-9)


def test_upload_and_allocation_fallbacks_do_not_change_the_registration(ctx, capi, monkeypatch):
    # MISLAM_PIN=1 (the context's pinned ring of rounds 3-5 instead of the runtime's pageable upload path, the default since round 6; read at context creation) in this process,
    # MISLAM_POOL=0 (plain hipMalloc / hipFree instead of the stream-ordered pool; decided once per process) in a child process: how the
    # bytes get there and where the buffers come from must not show in a single bit of the result
    import hashlib
    import os
    import subprocess
    import sys
    rng = np.random.default_rng(77)
    before = rng.uniform(-5, 5, (60000, 3)).astype(np.float32)                  # 720 KB per cloud: through the ring when it is on
    c, s = np.cos(0.1), np.sin(0.1)
    after = (before[rng.permutation(len(before))].astype(np.float64) @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]).T + 0.3).astype(np.float32)
    p = capi.icp_params(max_iterations=12)
    base = ctx.icp_register(before, after, p)

    def digest(r):
        return hashlib.sha256(np.asarray(r[0], np.float32).tobytes() + np.asarray(r[1], np.float32).tobytes() + np.int32(r[2]).tobytes() + np.float32(r[3]).tobytes()).hexdigest()

    monkeypatch.setenv("MISLAM_PIN", "1")
    with capi.Context(0) as c2:
        assert digest(c2.icp_register(before, after, p)) == digest(base)
    monkeypatch.delenv("MISLAM_PIN")
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from conftest import load_package\n"
        "capi = load_package().capi\n"
        "rng = np.random.default_rng(77)\n"
        "before = rng.uniform(-5, 5, (60000, 3)).astype(np.float32)\n"
        "c, s = np.cos(0.1), np.sin(0.1)\n"
        "after = (before[rng.permutation(len(before))].astype(np.float64) @ np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]).T + 0.3).astype(np.float32)\n"
        "with capi.Context(0) as ctx:\n"
        "    r = ctx.icp_register(before, after, capi.icp_params(max_iterations=12))\n"
        "print('DIGEST', hashlib.sha256(np.asarray(r[0], np.float32).tobytes() + np.asarray(r[1], np.float32).tobytes() + np.int32(r[2]).tobytes() + np.float32(r[3]).tobytes()).hexdigest())\n"
    ) % os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MISLAM_POOL="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert ("DIGEST " + digest(base)) in out.stdout


def test_search_phase_counters_of_the_counting_build(ctx, capi):
    # Round 6: mi_profile_search_phases -- the loop trip counts the measured instruction budget of the search kernel is built from
    # (profiles/r06_search_budget.md).  The counting build runs the same control flow: its registration is the product build's bit for bit, and the
    # counters obey the kernel's own structure.
    from conftest import synth_cloud
    before, after = synth_cloud(200000, seed=9)[:2]
    p = capi.icp_params(eps=0.0, max_iterations=-1, sync_every=4)
    ctx.icp_load(before, after, p)
    ctx.icp_run(4)
    ctx.search_stats(True)
    ctx.icp_run(4)
    ph = ctx.search_phases()
    cand, rows, hard, pts, nodes, leaves, wwaves, longest = ctx.search_stats(False)
    with pytest.raises(capi.MiSlamError):
        ctx.search_phases()                                   # counting is off again
    ctx.icp_run(4)
    counted = ctx.icp_result()
    ctx.icp_load(before, after, p)
    ctx.icp_run(12)
    plain = ctx.icp_result()
    assert np.array_equal(counted[0], plain[0]) and np.array_equal(counted[1], plain[1]) and counted[3] == plain[3]
    waves_per_launch = (len(before) + 63) // 64
    assert ph["waves"] == 4 * waves_per_launch and pts == 4 * len(before)
    assert ph["scan_waves"] + ph["walk_only_waves"] == ph["waves"] and ph["block_batches"] == ph["scan_waves"]
    assert 0 < ph["block_dealt"] <= ph["block_batches"] and ph["block_deal_passes"] >= ph["block_dealt"] and ph["block_deal_writes"] >= ph["block_dealt"]
    assert ph["rest_dealt"] <= ph["rest_rounds"] + ph["rest_batches4"] and ph["rest_waves"] <= ph["scan_waves"]     # (below 900 000 points the leftover rows go four per lane and batch)
    assert ph["walk_leaf_offers"] <= ph["walk_leaf_hits"] <= leaves <= ph["walk_leaf_children"] and ph["walk_only_waves"] <= wwaves <= ph["waves"]
    assert nodes >= wwaves > 0 and cand > 0 and rows > 0 and longest > 0
