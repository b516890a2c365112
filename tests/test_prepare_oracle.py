"""CPU suite: the input-stage restatement (oracle/prep_oracle.c) against the fixture generated from the reference's own
GetCloudsFromConfig stages (tests/golden/bunny_prepare.npz, oracle/make_golden_prepare.py) and against that code run live."""
import numpy as np
import pytest

from conftest import prepare_cases, prepare_kwargs


def test_restatement_reproduces_the_reference_fixture_bit_for_bit(oracle, golden):
    cases = prepare_cases(golden)
    assert len(cases) == 2
    for opt, raw, draws, want in cases:
        for side in "ba":
            got = oracle.prepare_cloud(raw, **prepare_kwargs(opt, draws, side))
            assert got.shape == want[side].shape
            assert np.array_equal(got.view(np.uint32), want[side].view(np.uint32)), (opt["seed"], side)


def test_stages_one_by_one(oracle):
    rng = np.random.default_rng(3)
    raw = rng.normal(size=(500, 3)).astype(np.float32) * np.float32(4.0) + np.float32(20.0)
    # nothing to do: a copy
    assert np.array_equal(oracle.prepare_cloud(raw), raw)
    # normalisation: the largest span becomes the spread, the centre of mass stays
    out = oracle.prepare_cloud(raw, spread=10.0)
    span = (out.max(0) - out.min(0)).max()
    assert abs(span - 10.0) < 1e-4 and np.abs(out.mean(0) - raw.mean(0)).max() < 1e-3
    # a cloud of identical points has no span: returned unchanged (common.cpp:88-89)
    same = np.repeat(raw[:1], 50, 0)
    assert np.array_equal(oracle.prepare_cloud(same, spread=10.0), same)
    # subcloud then shuffle: row i = raw[sub[shuffle[i]]]
    sub, shuf = rng.permutation(500)[:100].astype(np.int32), rng.permutation(100).astype(np.int32)
    assert np.array_equal(oracle.prepare_cloud(raw, subcloud_idx=sub, shuffle_idx=shuf), raw[sub[shuf]])
    # noise: only the flagged rows move, by at most spread * intensity per axis
    rows = np.sort(rng.permutation(500)[:60]).astype(np.int32)
    unit = rng.uniform(0, 1, (60, 3)).astype(np.float32)
    out = oracle.prepare_cloud(raw, noise_rows=rows, noise_unit=unit, noise_intensity=0.05)
    moved = np.abs(out - raw).max(1)
    reach = (raw.max(0) - raw.min(0)).max() * 0.05
    assert np.all(moved[np.setdiff1d(np.arange(500), rows)] == 0) and moved[rows].max() <= reach * 1.0001 and moved[rows].min() > 0
    # outliers land inside the cloud's bounds, after the cloud
    out = oracle.prepare_cloud(raw, outlier_unit=rng.uniform(0, 1, (9, 3)).astype(np.float32))
    assert out.shape == (509, 3) and np.array_equal(out[:500], raw)
    assert np.all(out[500:] >= raw.min(0)) and np.all(out[500:] <= raw.max(0))
    # the known transformation
    R = np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32)
    t = np.array([1, 2, 3], np.float32)
    assert np.abs(oracle.prepare_cloud(raw, R=R, t=t) - (raw @ R.T + t)).max() < 1e-4


@pytest.mark.parametrize("seed,kw", [
    (1, dict(resize_before=700, resize_after=900, noise_before=(0.5, 0.1), noise_after=(0.2, 0.01), outliers_before=3, outliers_after=0)),
    (77, dict(resize_before=None, resize_after=1500, noise_before=None, noise_after=(0.05, 0.3), outliers_before=0, outliers_after=40)),
    (4242, dict(resize_before=5000, resize_after=None, noise_before=(0.0, 0.5), noise_after=(1.0, 0.02), outliers_before=1, outliers_after=1)),
])
@pytest.mark.parametrize("spread", [None, 10.0, 1.0])
def test_restatement_vs_reference_live(oracle, ref, bunny, seed, kw, spread):
    # the reference's own code (oracle/_ref) on its own generators vs the restatement fed the outcomes derived from them
    raw = bunny[1][:2000]
    R = np.array([[0.6, -0.8, 0.0], [0.8, 0.6, 0.0], [0.0, 0.0, 1.0]], np.float32)
    t = np.array([0.5, -3.0, 2.0], np.float32)
    rb, ra = ref.clouds_from_config_full(raw, None, seed, R, t, spread=spread, **kw)
    db, da = ref.config_draws(len(raw), len(raw), seed, **kw)
    opt = dict(spread=spread, noise_before=kw["noise_before"], noise_after=kw["noise_after"], R=R, t=t)
    gb = oracle.prepare_cloud(raw, **prepare_kwargs(opt, {"b": db, "a": da}, "b"))
    ga = oracle.prepare_cloud(raw, **prepare_kwargs(opt, {"b": db, "a": da}, "a"))
    assert np.array_equal(gb.view(np.uint32), rb.view(np.uint32)) and np.array_equal(ga.view(np.uint32), ra.view(np.uint32))


def test_two_different_files(oracle, ref, bunny):
    # before and after loaded from different files (common.cpp:141-142), different sizes
    rawb, rawa = bunny[0][:1200], bunny[1][:1700]
    R, t = np.eye(3, dtype=np.float32), np.zeros(3, np.float32)
    rb, ra = ref.clouds_from_config_full(rawb, rawa, 9, R, t, spread=5.0, noise_before=(0.1, 0.1))
    db, da = ref.config_draws(len(rawb), len(rawa), 9, noise_before=(0.1, 0.1))
    opt = dict(spread=5.0, noise_before=(0.1, 0.1), noise_after=None, R=R, t=t)
    assert np.array_equal(oracle.prepare_cloud(rawb, **prepare_kwargs(opt, {"b": db, "a": da}, "b")), rb)
    assert np.array_equal(oracle.prepare_cloud(rawa, **prepare_kwargs(opt, {"b": db, "a": da}, "a")), ra)
