"""CPU suite: the C-ABI library loads and exports every entry point include/mi_slam.h declares; the pure host-side
helpers behave; without a device the compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "mi_slam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(capi):
    lib = capi.lib()
    names = header_functions()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert missing == []
    assert sorted(capi.EXPORTS) == names          # the binding's list is the header's list
    assert lib.mi_abi_version() == 4


def test_struct_layouts_match_header(capi):
    # mi_icp_params: 9 scalars + 7 reserved ints; mi_cpd_params: 8 scalars + 8 reserved ints (all 4-byte fields)
    assert C.sizeof(capi.IcpParams) == 16 * 4
    assert C.sizeof(capi.CpdParams) == 16 * 4
    p = capi.icp_params()
    assert (p.eps, p.max_iterations, p.max_distance_squared) == (pytest.approx(1e-3), -1, 1000.0)
    assert (p.dist_mode, p.compose_mode, p.filter_pairs, p.abort_on_increase) == (0, 0, 1, 0)     # cpu-slam rules
    q = capi.icp_params(cuda_slam=True)
    assert (q.dist_mode, q.compose_mode, q.filter_pairs, q.abort_on_increase) == (1, 1, 0, 1)     # cuda-slam rules
    c = capi.cpd_params()
    assert (c.weight, c.const_scale, c.max_iterations) == (pytest.approx(0.3), 0, -1)            # parser defaults


def test_shard_ranges_partition_the_target(capi):
    for m in (1, 7, 8, 14904, 10 ** 6, 10 ** 7 + 3):
        for world in (1, 2, 3, 4, 8):
            if m < world:
                continue
            edges = [capi.shard_range(m, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == m
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(capi.MiSlamError):
        capi.shard_range(10, 3, 2)


def test_source_shares_cover_the_moving_cloud(capi):
    for n in (1, 7, 255, 256, 511, 512, 1000, 14904, 10 ** 6, 10 ** 7 + 3):
        for world in (1, 2, 3, 4, 8):
            shares = [capi.source_share(n, r, world) for r in range(world)]
            assert sum(shares) == n and min(shares) >= 0
            if n >= 256 * world:                                  # dealt 64-point chunks: every rank whole chunks, but for the last one
                assert sum(1 for v in shares if v % 64) <= 1 and max(shares) - min(shares) <= 64
            else:
                assert max(shares) - min(shares) <= 1
    with pytest.raises(capi.MiSlamError):
        capi.source_share(10, 2, 2)


def test_key_packing_orders_like_distance_then_index(capi):
    rng = np.random.default_rng(0)
    d = np.concatenate([rng.uniform(0, 100, 200), [0.0, 0.0, 1e-30, 3.4e38, np.inf]]).astype(np.float32)
    i = rng.integers(0, 2 ** 31 - 1, len(d))
    keys = [capi.pack_key(x, y) for x, y in zip(d, i)]
    order = sorted(range(len(d)), key=lambda k: keys[k])
    ref_order = sorted(range(len(d)), key=lambda k: (float(d[k]), int(i[k])))
    assert order == ref_order
    for k, x, y in zip(keys, d, i):
        dd, ii = capi.unpack_key(k)
        assert np.float32(dd) == x or (np.isinf(x) and np.isinf(dd))
        assert ii == y


def test_no_cpu_fallback_without_a_device(capi):
    try:
        n = capi.device_count()
    except capi.MiSlamError as e:
        assert "no usable HIP device" in str(e) and "no CPU fallback" in str(e)
        with pytest.raises(capi.MiSlamError):
            capi.Context(0)
        return
    assert n >= 1          # on the GPU box the same call simply succeeds


def test_auto_batch_is_a_function_of_global_sizes_only(capi):
    # Every batch of a multi-GPU registration ends in a collective, so all ranks must pick the same batch.  A rank's own share of a
    # dealt moving cloud differs from its neighbours' by up to 64 points -- the batch therefore comes from the GLOBAL sizes
    # (mi_icp_auto_batch has no rank argument); the thresholds are crossed at the same total size on every rank count's shares.
    for world in (1, 2, 3, 8):
        for n_rank_at_threshold in (1.6e6, 3.6e6, 9.96e7):                  # 5e-11 s per point: 1e-4 s, 2e-4 s, 5e-3 s
            totals = [int(n_rank_at_threshold * world) + d for d in range(-200, 201, 40)]
            for n in totals:
                shares = [capi.source_share(n, r, world) for r in range(world)]
                assert max(shares) - min(shares) <= 64
                b = capi.icp_auto_batch(n, n, world, True, False)
                assert b in (1, 4, 8, 16)
            assert capi.icp_auto_batch(totals[0], totals[0], world, True, False) >= capi.icp_auto_batch(totals[-1], totals[-1], world, True, False)
    assert capi.icp_auto_batch(10 ** 6, 10 ** 6, 1, False, False) == 16 and capi.icp_auto_batch(10 ** 4, 10 ** 4, 1, False, True) == 16
    assert capi.icp_auto_batch(10 ** 6, 10 ** 6, 8, False, True) == 1            # every-pair search of 1e6 x 125 000 per rank: 18 ms
