"""GPU suite: context-level entry points -- eager loading of the device code (mi_ctx_preload)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_preload_is_idempotent_and_changes_no_result(capi, bunny):
    before, after = bunny
    p = capi.icp_params(max_iterations=5, max_distance_squared=400.0)
    with capi.Context(0) as lazy, capi.Context(0) as eager:
        eager.preload()
        eager.preload()
        a = lazy.icp_register(before, after, p)
        b = eager.icp_register(before, after, p)
        assert a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        ca = lazy.cpd_register(before[:2000], after[:2000], capi.cpd_params(max_iterations=3))
        cb = eager.cpd_register(before[:2000], after[:2000], capi.cpd_params(max_iterations=3))
        assert np.array_equal(ca[0], cb[0]) and np.array_equal(ca[1], cb[1])


def test_preload_at_creation_through_the_environment(capi, monkeypatch):
    monkeypatch.setenv("MISLAM_PRELOAD", "1")
    with capi.Context(0) as c:
        rng = np.random.default_rng(2)
        a = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
        assert c.icp_register(a, a + np.float32(0.05), capi.icp_params(eps=0.0, max_iterations=2))[2] == 2
