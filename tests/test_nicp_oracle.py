"""CPU suite: the non-iterative-registration restatement (oracle/nicp_oracle.c) against the fixtures generated from the reference's
own CPU build (tests/golden/bunny_nicp*, oracle/make_golden_nicp.py) and against that build run live where it is present."""
import numpy as np
import pytest

from conftest import frob


def rigid_pair(seed, n, noise=0.0):
    """Two clouds in the SAME point order related by a rigid motion -- the input the method is made for."""
    rng = np.random.default_rng(seed)
    b = (rng.normal(size=(n, 3)) * np.array([3.0, 1.5, 0.7])).astype(np.float32)          # distinct principal axes
    ang = 0.7
    ax = np.array([1.0, 2.0, 3.0]) / np.sqrt(14.0)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
    t = np.array([0.5, -1.0, 2.0])
    a = (b @ R.T + t + rng.normal(size=(n, 3)) * noise).astype(np.float32)
    return b, a, R, t


# ---------------------------------------------------------------------------------------------------------------------
# against the committed fixtures
# ---------------------------------------------------------------------------------------------------------------------
def test_bunny_single_results(oracle, golden, bunny):
    before, after = bunny
    g = golden.json("bunny_nicp.json")
    perms = golden.npz("bunny_nicp_perms.npz")["perms"].astype(np.int32)
    for k, s in enumerate(g["singles"]):
        R, t, e = oracle.nicp_single(before[perms[k]], after[perms[k]])
        assert frob(R, t, np.array(s["R"]), np.array(s["t"])) < 2e-5
        assert abs(e - s["approximated_error"]) < 2e-6 * s["approximated_error"]
    # the three permutations give different sign patterns of the same principal axes
    Rs = [np.array(s["R"]) for s in g["singles"]]
    assert all(abs(abs(np.linalg.det(R)) - 1) < 1e-4 for R in Rs)


@pytest.mark.parametrize("name,approx", [("none", 0), ("full", 1), ("hybrid", 2)])
def test_bunny_driver(oracle, golden, bunny, name, approx):
    before, after = bunny
    g = golden.json("bunny_nicp.json")
    z = golden.npz("bunny_nicp_perms.npz")
    perms, sub = z["perms"].astype(np.int32), z["sub_perm"].astype(np.int32)
    assert perms[:, :3].tolist() == g["order_heads"] and sub.tolist() == g["subcloud_idx"]
    r = g["runs"][name]
    R, t, reps, err = oracle.nicp(before, after, perms, sub, g["eps"], g["repetitions"], approx)
    assert reps == r["repetitions"]
    assert frob(R, t, np.array(r["R"]), np.array(r["t"])) < 2e-5
    assert abs(err - r["error"]) < 1e-5 * r["error"]


# ---------------------------------------------------------------------------------------------------------------------
# against the reference build run live
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed,n", [(0, 5), (1, 7), (2, 50), (3, 1000), (4, 4000)])
def test_single_matches_reference(oracle, ref, seed, n):
    rng = np.random.default_rng(seed)
    b = (rng.normal(size=(n, 3)) * rng.uniform(0.5, 3, 3)).astype(np.float32)
    a = (rng.normal(size=(n, 3)) * rng.uniform(0.5, 3, 3) + rng.normal(size=3)).astype(np.float32)
    Rr, tr, er = ref.nicp_single(b, a)
    Ro, to, eo = oracle.nicp_single(b, a)
    assert frob(Ro, to, Rr, tr) < 2e-5
    assert abs(eo - er) < 5e-6 * er


def test_single_recovers_a_rigid_motion_up_to_axis_signs(oracle, ref):
    # same point order in both clouds: the rotation is right whenever the two SVDs pick the same signs; the sign pattern
    # (one of 4 proper + 4 improper combinations) depends on the first points only
    b, a, R, t, = rigid_pair(5, 3000)
    seen = set()
    for k in range(12):
        p = np.random.default_rng(100 + k).permutation(len(b))
        Ro, to, eo = oracle.nicp_single(b[p], a[p])
        Rr, tr, er = ref.nicp_single(b[p], a[p])
        assert frob(Ro, to, Rr, tr) < 2e-5
        seen.add(tuple(np.sign(np.round(np.diag(R.T @ Ro), 3)).astype(int)))      # which axes came out flipped
        if eo < 1e-6:
            assert np.abs(Ro - R).max() < 1e-4 and np.abs(to - t).max() < 1e-3
    assert len(seen) >= 2                                                          # the permutations do change the signs


@pytest.mark.parametrize("approx", [0, 1, 2])
@pytest.mark.parametrize("seed", [666, 7])
def test_driver_matches_reference(oracle, ref, approx, seed):
    b, a, R, t = rigid_pair(seed, 2500, noise=0.01)
    reps, sub_n = 10, 300
    sub = ref.random_permutation(seed, len(b), 0)[:sub_n]
    perms = np.stack([ref.random_permutation(seed, len(b), 1 + k) for k in range(reps)])
    Rr, tr, nr, er = ref.nicp(b, a, 1e-3, reps, approx, False, sub_n, seed)
    Ro, to, no, eo = oracle.nicp(b, a, perms, sub, 1e-3, reps, approx)
    assert no == nr
    assert frob(Ro, to, Rr, tr) < 2e-5
    assert abs(eo - er) < 1e-5 * max(er, 1e-6) + 1e-9


def test_driver_whole_cloud_subcloud_draws_no_permutation(oracle, ref):
    # subcloudSize >= cloud size: GetSubcloud returns the cloud and leaves the generator alone (common.cpp:27-28)
    b, a, R, t = rigid_pair(3, 400, noise=0.02)
    reps, seed = 6, 21
    perms = np.stack([ref.random_permutation(seed, len(b), k) for k in range(reps)])
    Rr, tr, nr, er = ref.nicp(b, a, 1e-9, reps, 0, False, 1000, seed)
    Ro, to, no, eo = oracle.nicp(b, a, perms, None, 1e-9, reps, 0)
    assert no == nr and frob(Ro, to, Rr, tr) < 2e-5 and abs(eo - er) < 1e-5 * er
