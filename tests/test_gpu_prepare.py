"""GPU suite: the device input stage (mi_prepare_cloud) against the reference's own prepared clouds (fixture) and
against the CPU restatement on seeded inputs -- bit for bit: every stage is fp32 arithmetic in the reference's order."""
import numpy as np
import pytest

from conftest import prepare_cases, prepare_kwargs

pytestmark = pytest.mark.gpu


def same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_reference_fixture_bit_for_bit(ctx, golden):
    for opt, raw, draws, want in prepare_cases(golden):
        for side in "ba":
            got = ctx.prepare_cloud(raw, **prepare_kwargs(opt, draws, side))
            assert same_bits(got, want[side]), (opt["seed"], side, np.abs(got - want[side]).max())


def random_case(rng, n_raw, sub, noise, outliers, spread, move):
    raw = (rng.normal(size=(n_raw, 3)) * np.array([3.0, 1.0, 0.2]) + np.array([50.0, -7.0, 0.3])).astype(np.float32)
    kw = dict(spread=spread)
    n = n_raw
    if sub:
        n = max(1, n_raw // 3)
        kw["subcloud_idx"] = rng.permutation(n_raw)[:n].astype(np.int32)
    kw["shuffle_idx"] = rng.permutation(n).astype(np.int32)
    if noise:
        k = max(1, n // 4)
        kw.update(noise_rows=np.sort(rng.permutation(n)[:k]).astype(np.int32), noise_unit=rng.uniform(0, 1, (k, 3)).astype(np.float32),
                  noise_intensity=0.07)
    if outliers:
        kw["outlier_unit"] = rng.uniform(0, 1, (outliers, 3)).astype(np.float32)
    if move:
        kw.update(R=np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32), t=np.array([10, -20, 30], np.float32))
    return raw, kw


@pytest.mark.parametrize("n_raw", [1, 2, 63, 64, 65, 1000, 4097, 100003])
def test_matches_the_restatement_on_seeded_inputs(ctx, oracle, n_raw):
    rng = np.random.default_rng(n_raw)
    for sub in (False, True):
        for noise, outliers, spread, move in ((False, 0, None, False), (True, 5, 10.0, True), (False, 0, 10.0, False),
                                              (True, 0, None, True), (False, 17, 1.0, True)):
            raw, kw = random_case(rng, n_raw, sub, noise, outliers, spread, move)
            got = ctx.prepare_cloud(raw, **kw)
            want = oracle.prepare_cloud(raw, **kw)
            assert same_bits(got, want), (n_raw, sub, noise, outliers, spread, move, np.abs(got - want).max())


def test_million_points_sequential_centre(ctx, oracle):
    # the centre of mass is a sequential fp32 sum in the reference: at 1e6 points with an offset it is far from the exact mean,
    # and the normalised cloud inherits that -- the device reproduces those very roundings
    rng = np.random.default_rng(8)
    raw = (rng.uniform(-5, 5, (1000000, 3)) + np.array([100.0, 0.0, -30.0])).astype(np.float32)
    shuf = rng.permutation(len(raw)).astype(np.int32)
    got = ctx.prepare_cloud(raw, shuffle_idx=shuf, spread=10.0)
    want = oracle.prepare_cloud(raw, shuffle_idx=shuf, spread=10.0)
    assert same_bits(got, want)
    # and those roundings are visible: the reference's centre is not the exact mean of this cloud
    shift = (got.astype(np.float64).mean(0) - raw.astype(np.float64).mean(0))
    print("centre of the normalised cloud - exact centre of the raw cloud:", shift)


def test_degenerate_and_argument_checks(ctx, capi, oracle):
    same = np.tile(np.array([[1.5, -2.0, 3.25]], np.float32), (300, 1))
    assert same_bits(ctx.prepare_cloud(same, spread=10.0), same)          # no span: returned unchanged (common.cpp:88-89)
    raw = np.arange(30, dtype=np.float32).reshape(10, 3)
    with pytest.raises(capi.MiSlamError):
        ctx.prepare_cloud(raw, subcloud_idx=np.array([0, 10], np.int32))                 # index outside the raw cloud
    with pytest.raises(capi.MiSlamError):
        ctx.prepare_cloud(raw, shuffle_idx=np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 10], np.int32))
    with pytest.raises(capi.MiSlamError):
        ctx.prepare_cloud(raw, noise_rows=np.array([3, 2], np.int32), noise_unit=np.zeros((2, 3), np.float32))   # not ascending
    with pytest.raises(capi.MiSlamError):
        ctx.prepare_cloud(raw[:0])


def test_prepared_clouds_register(ctx, capi, golden):
    # end to end: prepare both clouds of the fixture's second case on the device, then ICP recovers the known transformation
    opt, raw, draws, want = prepare_cases(golden)[1]
    before = ctx.prepare_cloud(raw, **prepare_kwargs(opt, draws, "b"))
    after = ctx.prepare_cloud(raw, **dict(prepare_kwargs(opt, draws, "a"), noise_rows=None, noise_unit=None, outlier_unit=None))
    R, t, it, err = ctx.icp_register(before, after, capi.icp_params(eps=1e-6, max_iterations=80, max_distance_squared=1e9))
    assert np.abs(R - np.array(opt["R"])).max() < 1e-3 and np.abs(t - np.array(opt["t"])).max() < 1e-2
