"""GPU suite: the correspondence search (K1 every pair, K1t box hierarchy, K1g cell grid) through the C ABI against the oracle --
indices and distance bits must be IDENTICAL (strict '<', lowest index wins ties), in both distance arithmetics."""
import numpy as np
import pytest

from conftest import synth_cloud

pytestmark = pytest.mark.gpu


def check(ctx, capi, oracle, src, tgt, mode):
    # every execution strategy -- every pair (K1), the exact box hierarchy (K1t), the cell grid with its hierarchy fallback
    # (K1g) -- must reproduce the oracle bit for bit
    ridx, rd2 = oracle.nn_search(src, tgt, dist_mode=mode)
    for nn_mode in (capi.NN_BRUTEFORCE, capi.NN_TREE, capi.NN_GRID):
        idx, d2 = ctx.nn_search(src, tgt, mode, nn_mode)
        assert np.array_equal(idx, ridx), "nn_mode %d" % nn_mode
        assert np.array_equal(d2.view(np.uint32), rd2.view(np.uint32)), "nn_mode %d" % nn_mode


def test_correspondences_known_answer(ctx, capi):
    # CorrespondencesTest, source/cuda-slam/cudacommon.cu:291-317
    n = 100
    src = np.repeat(np.arange(n, dtype=np.float32)[:, None], 3, axis=1)
    tgt = src[::-1].copy()
    idx, d2 = ctx.nn_search(src, tgt)
    assert np.array_equal(idx, n - 1 - np.arange(n))
    assert np.all(d2 == 0)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,m", [(1, 1), (1, 17), (63, 5), (64, 16), (257, 1000), (1000, 333), (5000, 4999), (20011, 19997)])
def test_random_clouds_bit_exact(ctx, capi, oracle, n, m, mode):
    rng = np.random.default_rng(n * 7919 + m)
    src = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    tgt = rng.uniform(-5, 5, (m, 3)).astype(np.float32)
    check(ctx, capi, oracle, src, tgt, mode)


@pytest.mark.parametrize("mode", [0, 1])
def test_ties_resolve_to_lowest_index(ctx, capi, oracle, mode):
    rng = np.random.default_rng(5)
    base = rng.uniform(-5, 5, (3000, 3)).astype(np.float32)
    tgt = np.concatenate([base, base[::-1], base])           # every target three times, scattered over all chunks
    src = np.concatenate([base[:500], base[1000:1500] + np.float32(1e-3), rng.uniform(-5, 5, (777, 3)).astype(np.float32)])
    check(ctx, capi, oracle, src, tgt, mode)
    idx, _ = ctx.nn_search(src[:500], tgt, mode)
    assert np.array_equal(idx, np.arange(500))                 # exact hits: first copy wins


def test_adversarial_order_descending_distance(ctx, capi, oracle):
    # targets sorted so that every block improves the running minimum: the re-scan path runs on every block
    rng = np.random.default_rng(9)
    tgt = rng.uniform(-5, 5, (4096, 3)).astype(np.float32)
    src = np.zeros((300, 3), np.float32) + rng.normal(scale=1e-3, size=(300, 3)).astype(np.float32)
    order = np.argsort(-(tgt.astype(np.float64) ** 2).sum(1))
    check(ctx, capi, oracle, src, tgt[order].copy(), 0)


@pytest.mark.parametrize("nn_mode", [1, 2, 3])
def test_bunny_iter0_matches_golden(ctx, capi, golden, bunny, nn_mode):
    # bunny has every vertex ~6 times (face-corner expansion): exact ties everywhere, the lowest index must win
    before, after = bunny
    g = golden.npz("bunny_icp_iter0.npz")
    idx, d2 = ctx.nn_search(before, after, 0, nn_mode)
    keep = d2 < np.float32(400.0)
    assert np.array_equal(np.nonzero(keep)[0], g["idx_before"])
    assert np.array_equal(idx[keep], g["idx_after"])


@pytest.mark.parametrize("mode", [0, 1])
def test_tree_equals_bruteforce_on_clustered_and_degenerate_clouds(ctx, capi, mode):
    rng = np.random.default_rng(77)
    # clustered targets far from the sources, a plane, a line, all-identical points, one outlier at 1e6
    clouds = [
        np.concatenate([rng.normal(loc=c, scale=0.05, size=(3000, 3)) for c in ((0, 0, 0), (10, 10, 10), (-7, 3, 1))]),
        np.stack([rng.uniform(-5, 5, 9000), rng.uniform(-5, 5, 9000), np.zeros(9000)], 1),
        np.stack([np.linspace(-3, 3, 5000), np.zeros(5000), np.zeros(5000)], 1),
        np.tile(np.array([[1.5, -2.5, 0.25]]), (4096, 1)),
        np.concatenate([rng.uniform(-1, 1, (5000, 3)), [[1e6, -1e6, 1e6]]]),
    ]
    src = np.concatenate([rng.uniform(-12, 12, (3000, 3)), rng.normal(size=(1000, 3)) * 1e-3, [[1e6, -1e6, 1e6]]]).astype(np.float32)
    for tgt in clouds:
        tgt = tgt.astype(np.float32)
        a = ctx.nn_search(src, tgt, mode, capi.NN_BRUTEFORCE)
        for indexed in (capi.NN_TREE, capi.NN_GRID):
            b = ctx.nn_search(src, tgt, mode, indexed)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), indexed


@pytest.mark.parametrize("ppc", ["0.25", "1", "2", "3", "8", "16", "64"])   # 2-3: around the most trips one deal takes; 16: lanes with 32 trips (lockstep)
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("deal", ["0", "1"])                                  # leftover rows in batches of four per lane / dealt out one per lane
def test_grid_is_exact_at_every_cell_size(capi, oracle, monkeypatch, ppc, mode, deal):
    # the grid's cell size is a speed knob (MISLAM_GRID_PPC = mean points per cell, read at context creation): from cells far
    # smaller than the point spacing (most rows empty, radii of many cells -> hierarchy fallback for the far queries) to cells so
    # crowded that the candidate budget sends lanes to the hierarchy mid-scan -- the answer must not change by a bit
    monkeypatch.setenv("MISLAM_GRID_PPC", ppc)
    monkeypatch.setenv("MISLAM_GRID_DEAL_ROWS", deal)
    rng = np.random.default_rng(31)
    base = rng.uniform(-5, 5, (6000, 3)).astype(np.float32)
    tgt = np.concatenate([base, base[:2000]])                       # duplicates: ties
    src = np.concatenate([base[100:900], (base[:3000] + rng.normal(scale=0.05, size=(3000, 3))).astype(np.float32),
                          rng.uniform(-20, 20, (333, 3)).astype(np.float32)])
    ridx, rd2 = oracle.nn_search(src, tgt, dist_mode=mode)
    with capi.Context(0) as c2:
        idx, d2 = c2.nn_search(src, tgt, mode, capi.NN_GRID)
    assert np.array_equal(idx, ridx) and np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))


@pytest.mark.parametrize("n", [30000, 1000000])
def test_leftover_rows_dealt_or_in_batches_same_bits(capi, monkeypatch, n):
    # K1g takes a wave's leftover cell rows four per lane and batch below 900 000 moving points and deals them out one per lane above
    # (MISLAM_GRID_DEAL_ROWS=0 / 1 forces either, read at context creation): the plain search and a fused ICP run must not change by a bit
    from conftest import synth_cloud
    before, after = synth_cloud(n, seed=5)[:2]
    out = []
    for deal in ("0", "1"):
        monkeypatch.setenv("MISLAM_GRID_DEAL_ROWS", deal)
        with capi.Context(0) as c2:
            idx, d2 = c2.nn_search(before, after, 0, capi.NN_GRID)
            reg = c2.icp_register(before, after, capi.icp_params(eps=0.0, max_iterations=6))
            out.append((idx, d2.view(np.uint32), reg))
    (i0, d0, r0), (i1, d1, r1) = out
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    assert r0[2] == r1[2] == 6 and np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1]) and r0[3] == r1[3]


@pytest.mark.parametrize("mode", [0, 1])
def test_grid_boundary_and_outside_queries(ctx, capi, oracle, mode):
    # queries exactly on cell faces and corners, on the bounding box, just outside it and far outside it; fixed points on a
    # lattice that coincides with cell faces (every rounding slack of nn_grid.hip is exercised)
    g = np.arange(16, dtype=np.float32)
    tgt = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)          # 4096 lattice points, ext 15 per axis
    rng = np.random.default_rng(3)
    src = np.concatenate([tgt[::7] + np.float32(0.5), tgt[::5], tgt[::11] + np.float32([0.5, 0, 0]),
                          rng.uniform(-1, 16, (2000, 3)).astype(np.float32), rng.uniform(-40, 60, (300, 3)).astype(np.float32),
                          np.float32([[-1e-6, 7.5, 7.5], [15.000001, 15, 15], [7.5, 7.5, 7.5], [1e4, 1e4, 1e4]])]).astype(np.float32)
    ridx, rd2 = oracle.nn_search(src, tgt, dist_mode=mode)
    idx, d2 = ctx.nn_search(src, tgt, mode, capi.NN_GRID)
    assert np.array_equal(idx, ridx) and np.array_equal(d2.view(np.uint32), rd2.view(np.uint32))


@pytest.mark.parametrize("ppc", ["0.5", "1.25", "4"])
@pytest.mark.parametrize("mode", [0, 1])
def test_grid_serves_queries_outside_the_cloud_exactly(capi, monkeypatch, ppc, mode):
    # Round 5: a query OUTSIDE the grid's extent by g cells is served by the scan as long as the lens its sphere cuts out of the cloud stays within
    # two cells of its clamped own cell (r^2 <= (g_a + 2)^2 + the other axes' gaps^2, nn_grid.hip grid_lane_cap2) instead of walking the hierarchy
    # as soon as its neighbour is farther than two cells.  Shells of queries at 0.3 .. 40 cells outside a uniform cloud -- off faces, edges and
    # corners, axis by axis and sign by sign -- plus queries whose lens is too wide (sparse corner regions) and that must still fall back: bit for bit
    # the every-pair search, with and without a starting candidate (plain search: none; fused ICP: the previous match).
    monkeypatch.setenv("MISLAM_GRID_PPC", ppc)
    rng = np.random.default_rng(77)
    tgt = rng.uniform(-5, 5, (24000, 3)).astype(np.float32)
    tgt = tgt[~((tgt[:, 0] > 3) & (tgt[:, 1] > 3) & (tgt[:, 2] > 3))]             # a corner bitten off: outside queries there see a hollow
    h = 10.0 / (len(tgt) / float(ppc)) ** (1.0 / 3.0)                            # ~ the cell size the grid will choose
    parts = []
    for d_cells in (0.3, 1.0, 1.9, 2.1, 3.0, 4.5, 7.0, 8.5, 12.0, 15.9, 16.1, 25.0, 40.0):
        for axes in ((0,), (1,), (2,), (0, 1), (1, 2), (0, 2), (0, 1, 2)):
            q = rng.uniform(-5, 5, (40, 3))
            for a in axes:
                sign = rng.choice([-1.0, 1.0], 40)
                q[:, a] = sign * (5.0 + d_cells * h * rng.uniform(0.9, 1.1, 40))
            parts.append(q)
    src = np.concatenate(parts + [rng.uniform(-6, 6, (3000, 3))]).astype(np.float32)
    with capi.Context(0) as c2:
        want = c2.nn_search(src, tgt, mode, capi.NN_BRUTEFORCE)
        got = c2.nn_search(src, tgt, mode, capi.NN_GRID)
        assert np.array_equal(want[0], got[0]) and np.array_equal(want[1].view(np.uint32), got[1].view(np.uint32))
        # the fused iteration: starting candidates = previous matches, the moving cloud drifting across the faces of the fixed one
        runs = [c2.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=5, nn_mode=nn, dist_mode=mode)) for nn in (capi.NN_BRUTEFORCE, capi.NN_GRID)]
        assert runs[0][2] == runs[1][2] and np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][3] == runs[1][3]


@pytest.mark.parametrize("mode", [0, 1])
def test_grid_on_awkward_coordinate_ranges(ctx, capi, mode):
    # coordinates far from the origin (the fp32 lattice is coarser than the cell size: duplicates, crowded cells), a thin slab with
    # a huge aspect ratio, a tiny cloud (extents ~1e-6), a line stretched over 1e6 with dense knots, two far-apart clusters: the
    # grid's plan, its slacks and the hierarchy fallback against the every-pair search, bit for bit
    rng = np.random.default_rng(2026)
    n = 20000
    cases = []
    cases.append(1.0e5 + rng.uniform(0, 1, (n, 3)))
    cases.append(np.stack([rng.uniform(0, 1000, n), rng.uniform(0, 1000, n), rng.uniform(0, 1e-3, n)], 1))
    cases.append(rng.uniform(-1, 1, (n, 3)) * 1e-6)
    knots = rng.uniform(0, 1e6, 40)
    cases.append(np.stack([knots[rng.integers(0, 40, n)] + rng.normal(scale=0.01, size=n), np.zeros(n), rng.normal(scale=1e-4, size=n)], 1))
    cases.append(np.concatenate([rng.normal(loc=0, scale=0.1, size=(n // 2, 3)), rng.normal(loc=5e3, scale=0.1, size=(n // 2, 3))]))
    for tgt in cases:
        tgt = tgt.astype(np.float32)
        lo, hi = tgt.min(0), tgt.max(0)
        ext = np.maximum(hi - lo, 1e-9)
        src = np.concatenate([tgt[rng.integers(0, len(tgt), 3000)] + (rng.normal(size=(3000, 3)) * ext * 1e-3).astype(np.float32),
                              (lo + rng.uniform(-0.3, 1.3, (3000, 3)) * ext).astype(np.float32),
                              tgt[:500]]).astype(np.float32)
        a = ctx.nn_search(src, tgt, mode, capi.NN_BRUTEFORCE)
        for indexed in (capi.NN_GRID, capi.NN_TREE):
            b = ctx.nn_search(src, tgt, mode, indexed)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), indexed
        # and through the fused ICP iteration (warm starts, match slots): same trajectory as the every-pair path
        runs = [ctx.icp_register(src, tgt, capi.icp_params(eps=0.0, max_iterations=3, dist_mode=mode, nn_mode=nn))
                for nn in (capi.NN_BRUTEFORCE, capi.NN_GRID)]
        assert runs[0][2] == runs[1][2] and np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])


def test_large_sampled_rows_and_properties(ctx, capi, oracle):
    # BASELINE size class (cfg 2, N = M = 1e5): full check is 1e10 pairs -- the oracle re-computes a sample of source rows;
    # the rest is covered by size-independent properties (self-search is the identity with d2 = 0; the reported d2 is the
    # true distance to the reported index; no sampled target is closer).
    before, after, _, _ = synth_cloud(100000)
    idx, d2 = ctx.nn_search(before, after, 0, capi.NN_GRID)
    bidx, bd2 = ctx.nn_search(before, after, 0, capi.NN_BRUTEFORCE)
    assert np.array_equal(idx, bidx) and np.array_equal(d2.view(np.uint32), bd2.view(np.uint32))   # all 1e10 pairs vs the grid
    tidx, td2 = ctx.nn_search(before, after, 0, capi.NN_TREE)
    assert np.array_equal(tidx, bidx) and np.array_equal(td2.view(np.uint32), bd2.view(np.uint32))
    rows = np.random.default_rng(0).choice(len(before), 256, replace=False)
    ridx, rd2 = oracle.nn_search(before[rows], after)
    assert np.array_equal(idx[rows], ridx)
    assert np.array_equal(d2[rows].view(np.uint32), rd2.view(np.uint32))
    diff = after[idx] - before
    assert np.array_equal(((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]).view(np.uint32),
                          d2.view(np.uint32))
    sidx, sd2 = ctx.nn_search(after, after)
    assert np.array_equal(sidx, np.arange(len(after))) and np.all(sd2 == 0)


def test_full_bench_size_properties(ctx, capi, oracle):
    # BASELINE.json's headline size (N = M = 1e6): the cell grid and the box hierarchy against ALL 1e12 pairs of the every-pair
    # kernel (which is pinned to the oracle above), sampled rows against the oracle itself, and the size-independent properties
    before, after, Rm, tm = synth_cloud(1000000)
    idx, d2 = ctx.nn_search(before, after, 0, capi.NN_GRID)
    bidx, bd2 = ctx.nn_search(before, after, 0, capi.NN_BRUTEFORCE)
    assert np.array_equal(idx, bidx) and np.array_equal(d2.view(np.uint32), bd2.view(np.uint32))
    tidx, td2 = ctx.nn_search(before, after, 0, capi.NN_TREE)
    assert np.array_equal(tidx, bidx) and np.array_equal(td2.view(np.uint32), bd2.view(np.uint32))
    rows = np.random.default_rng(1).choice(len(before), 64, replace=False)
    ridx, rd2 = oracle.nn_search(before[rows], after)
    assert np.array_equal(idx[rows], ridx) and np.array_equal(d2[rows].view(np.uint32), rd2.view(np.uint32))
    diff = after[idx] - before
    assert np.array_equal(((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]).view(np.uint32),
                          d2.view(np.uint32))
    # searching the registered cloud finds every point's own image: the permutation is recovered exactly
    moved = (before.astype(np.float64) @ Rm.astype(np.float64).T + tm.astype(np.float64)).astype(np.float32)
    pidx, pd2 = ctx.nn_search(moved, after)
    assert len(np.unique(pidx)) == len(after) and float(pd2.max()) < 1e-8
    sidx, sd2 = ctx.nn_search(after, after)
    assert np.array_equal(sidx, np.arange(len(after))) and np.all(sd2 == 0)


def test_ten_million_points(ctx, capi, oracle):
    # the largest size BASELINE.json lists (N = M = 1e7): sampled rows against the oracle, the distance really is the distance to
    # the reported index, and the registered cloud finds every point's own image
    before, after, Rm, tm = synth_cloud(10000000)
    idx, d2 = ctx.nn_search(before, after)
    rows = np.random.default_rng(2).choice(len(before), 32, replace=False)
    ridx, rd2 = oracle.nn_search(before[rows], after)
    assert np.array_equal(idx[rows], ridx) and np.array_equal(d2[rows].view(np.uint32), rd2.view(np.uint32))
    diff = after[idx] - before
    assert np.array_equal(((diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]).view(np.uint32),
                          d2.view(np.uint32))
    moved = (before.astype(np.float64) @ Rm.astype(np.float64).T + tm.astype(np.float64)).astype(np.float32)
    pidx, pd2 = ctx.nn_search(moved, after)
    assert np.array_equal(np.sort(pidx), np.arange(len(after))) and float(pd2.max()) < 1e-8


def test_rejects_bad_arguments(ctx, capi):
    with pytest.raises(capi.MiSlamError):
        ctx.nn_search(np.zeros((4, 3), np.float32), np.zeros((0, 3), np.float32))
    with pytest.raises(capi.MiSlamError):
        ctx.nn_search(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), dist_mode=7)
    idx, d2 = ctx.nn_search(np.zeros((0, 3), np.float32), np.zeros((4, 3), np.float32))     # empty source: nothing to do
    assert len(idx) == 0
