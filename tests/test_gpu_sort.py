"""The library's own device radix sort (radix_sort.hip) -- the Hilbert ordering of the index builds runs on it.  Bit-exact against
numpy's stable sort: sorted keys AND the order of equal keys."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 4096 * 64, 4096 * 64 + 1, 1_000_003])
@pytest.mark.parametrize("bits", [10, 20, 30])
def test_sort_matches_numpy_stable(ctx, n, bits):
    rng = np.random.default_rng(n * 31 + bits)
    keys = rng.integers(0, 1 << bits, size=n, dtype=np.uint32)
    vals = np.arange(n, dtype=np.int32)
    k, v = ctx.selftest_sort_pairs(keys, vals, bits)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k, keys[order])
    assert np.array_equal(v, order.astype(np.int32))


def test_sort_heavy_duplicates_and_presorted(ctx):
    rng = np.random.default_rng(5)
    n = 300_000
    for keys in (np.zeros(n, np.uint32), np.full(n, (1 << 30) - 1, np.uint32), rng.integers(0, 3, n).astype(np.uint32) << 17,
                 np.sort(rng.integers(0, 1 << 30, n).astype(np.uint32)), np.sort(rng.integers(0, 1 << 30, n).astype(np.uint32))[::-1]):
        vals = rng.permutation(n).astype(np.int32)
        k, v = ctx.selftest_sort_pairs(keys, vals, 30)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(k, keys[order]) and np.array_equal(v, vals[order])


def test_sort_ignores_bits_above_the_range(ctx):
    rng = np.random.default_rng(6)
    keys = rng.integers(0, 1 << 32, size=50_000, dtype=np.uint64).astype(np.uint32)
    vals = np.arange(keys.size, dtype=np.int32)
    k, v = ctx.selftest_sort_pairs(keys, vals, 20)
    order = np.argsort(keys & 0xFFFFF, kind="stable")
    assert np.array_equal(v, order.astype(np.int32)) and np.array_equal(k, keys[order])


def test_sort_rejects_bad_arguments(ctx, capi):
    with pytest.raises(capi.MiSlamError):
        ctx.selftest_sort_pairs(np.zeros(4, np.uint32), np.zeros(4, np.int32), 12)
