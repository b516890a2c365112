"""GPU suite: non-iterative registration ("method": "nicp") through the C ABI against the plain-C restatement
(oracle/nicp_oracle.c, pinned against the reference's CPU build by tests/test_nicp_oracle.py) and the fixtures generated from
that build (tests/golden/bunny_nicp*).

The product forms each candidate from one fused fp64 moments pass plus the first three points of the repetition's permutation;
the reference runs an fp32 SVD of the permuted 3 x N matrices and sums centroids and errors sequentially in fp32 in permuted
order.  Same candidates, same choice; R agrees to ~1e-5, and t and the errors carry the reference's own fp32 summation noise
(a few 1e-5 relative at 15 000 points -- DESIGN.md, deviation 1), hence the tolerances below."""
import numpy as np
import pytest

from conftest import frob
from test_nicp_oracle import rigid_pair

pytestmark = pytest.mark.gpu


def draws(seed, n, reps, sub_n):
    rng = np.random.default_rng(seed)
    sub = rng.permutation(n)[:sub_n].astype(np.int32) if sub_n < n else None
    perms = np.stack([rng.permutation(n) for _ in range(reps)]).astype(np.int32)
    return perms, sub


@pytest.mark.parametrize("name,approx", [("none", 0), ("full", 1), ("hybrid", 2)])
def test_bunny_matches_cpu_slam(ctx, capi, golden, bunny, name, approx):
    before, after = bunny
    g = golden.json("bunny_nicp.json")
    r = g["runs"][name]
    p = capi.nicp_params(eps=g["eps"], max_repetitions=g["repetitions"], approximation=approx)
    R, t, reps, err = ctx.nicp_register(before, after, p, np.array(g["order_heads"], np.int32), np.array(g["subcloud_idx"], np.int32))
    assert reps == r["repetitions"]
    assert frob(R, t, np.array(r["R"]), np.array(r["t"])) < 1e-4                  # north_star bar
    assert abs(err - r["error"]) < 1e-4 * r["error"]


@pytest.mark.parametrize("approx", [0, 1, 2])
@pytest.mark.parametrize("seed,n,sub_n,noise", [(1, 2500, 300, 0.01), (2, 40000, 1000, 0.0), (3, 64, 64, 0.05), (4, 4, 4, 0.0)])
def test_matches_oracle(ctx, capi, oracle, approx, seed, n, sub_n, noise):
    # (three points span a plane only: the third axis' sign is rounding noise in the reference itself -- four is the smallest
    # cloud with a defined answer)
    b, a, R0, t0 = rigid_pair(seed, n, noise)
    reps = 9
    perms, sub = draws(100 + seed, n, reps, sub_n)
    Ro, to, no, eo = oracle.nicp(b, a, perms, sub, 1e-4, reps, approx)
    p = capi.nicp_params(eps=1e-4, max_repetitions=reps, approximation=approx)
    R, t, nr, err = ctx.nicp_register(b, a, p, perms[:, :3], sub)
    assert nr == no
    assert frob(R, t, Ro, to) < 1e-4
    assert abs(err - eo) < 2e-4 * max(eo, 1e-4)


def test_finds_the_rigid_motion_and_stops_early(ctx, capi, oracle):
    # mode none: the first repetition whose subcloud error is <= eps ends the run (noniterative.cpp:238-242)
    b, a, R0, t0 = rigid_pair(9, 20000)
    reps = 32
    perms, sub = draws(5, len(b), reps, 1000)
    p = capi.nicp_params(eps=1e-5, max_repetitions=reps, approximation=0)
    R, t, nr, err = ctx.nicp_register(b, a, p, perms[:, :3], sub)
    Ro, to, no, eo = oracle.nicp(b, a, perms, sub, 1e-5, reps, 0)
    assert nr == no and nr < reps and err <= 1e-5
    assert np.abs(R - R0).max() < 1e-4 and np.abs(t - t0).max() < 1e-3
    assert frob(R, t, Ro, to) < 1e-4


def test_unequal_sizes_and_default_repetitions(ctx, capi, oracle):
    # |after| > |before|: the permutations cover min(m, n) points, the rest of the larger cloud stays in place (common.h:101-108);
    # max_repetitions = -1 means 20 (noniterative.cpp:207-208)
    b, a, R0, t0 = rigid_pair(6, 1800, 0.01)
    extra = (np.random.default_rng(0).normal(size=(200, 3)) * 2 + t0).astype(np.float32)
    a2 = np.concatenate([a, extra])
    perms, sub = draws(8, len(b), 20, 400)
    for approx in (0, 2):
        Ro, to, no, eo = oracle.nicp(b, a2, perms, sub, 1e-6, 20, approx)
        p = capi.nicp_params(eps=1e-6, max_repetitions=-1, approximation=approx)
        R, t, nr, err = ctx.nicp_register(b, a2, p, perms[:, :3], sub)
        assert nr == no == 20
        assert frob(R, t, Ro, to) < 1e-4 and abs(err - eo) < 2e-4 * eo


def test_large_cloud_uses_the_box_hierarchy_for_the_error(ctx, capi, oracle):
    # small subclouds are scored by the every-pair kernel; from 4e9 pairs on (here the whole 70 000-point cloud against itself)
    # the box hierarchy takes over: same exact neighbours, same error
    b, a, R0, t0 = rigid_pair(12, 70000, 0.002)
    perms, sub = draws(3, len(b), 2, len(b))
    assert sub is None
    p = capi.nicp_params(eps=1e-9, max_repetitions=2, approximation=0)
    R, t, nr, err = ctx.nicp_register(b, a, p, perms[:, :3], sub)
    Ro, to, no, eo = oracle.nicp(b, a, perms, sub, 1e-9, 2, 0)
    assert nr == no == 2 and frob(R, t, Ro, to) < 1e-4 and abs(err - eo) < 2e-4 * eo


def test_rejects_bad_arguments(ctx, capi):
    b, a, _, _ = rigid_pair(0, 100)
    perms, sub = draws(0, 100, 5, 20)
    ok = capi.nicp_params(max_repetitions=5, approximation=0)
    with pytest.raises(capi.MiSlamError):
        ctx.nicp_register(b[:2], a, ok, perms[:, :3], None)                    # fewer than 3 points
    with pytest.raises(capi.MiSlamError):
        ctx.nicp_register(b, a, capi.nicp_params(max_repetitions=5, approximation=7), perms[:, :3], sub)
    bad = perms[:, :3].copy()
    bad[2, 1] = 100
    with pytest.raises(capi.MiSlamError):
        ctx.nicp_register(b, a, ok, bad, sub)                                  # index outside the clouds
    with pytest.raises(capi.MiSlamError):
        ctx.nicp_register(b, a[:50], capi.nicp_params(max_repetitions=5, approximation=2), np.minimum(perms[:, :3], 49), sub)   # |before| > |after|
    with pytest.raises(ValueError):
        ctx.nicp_register(b, a, ok, perms[:3, :3], sub)                        # wrong number of heads
