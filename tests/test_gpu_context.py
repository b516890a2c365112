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


def test_contexts_in_turn_and_in_threads_with_changing_sizes(capi):
    # Round 4: each context retires its outgrown buffers on its OWN list (released behind a drain of its own stream), device memory
    # comes out of a private pool, and a load runs on three streams with two scratch sets.  Registrations of changing sizes, on two
    # contexts taken in turns and then from two host threads at once, must give the bits a fresh context gives for the same clouds.
    import threading
    from conftest import synth_cloud
    sizes = [12000, 150000, 30000, 400000, 20000, 90000]
    clouds = {n: synth_cloud(n, seed=n)[:2] for n in sizes}
    p = capi.icp_params(eps=0.0, max_iterations=4)
    want = {}
    for n in sizes:
        with capi.Context(0) as fresh:
            want[n] = fresh.icp_register(*clouds[n], p)

    def same(a, b):
        return a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[3] == b[3]

    with capi.Context(0) as c1, capi.Context(0) as c2:
        for k, n in enumerate(sizes + sizes[::-1]):                      # growing and shrinking, alternating contexts
            assert same((c1 if k % 2 == 0 else c2).icp_register(*clouds[n], p), want[n]), n
        bad = []

        def work(ctx, order):
            for n in order:
                if not same(ctx.icp_register(*clouds[n], p), want[n]):
                    bad.append(n)
        t1 = threading.Thread(target=work, args=(c1, sizes * 2))
        t2 = threading.Thread(target=work, args=(c2, sizes[::-1] * 2))
        t1.start(); t2.start(); t1.join(); t2.join()
        assert not bad, bad
        # a search primitive and a CPD call between registrations reuse the same workspace
        i1, d1 = c1.nn_search(clouds[12000][0], clouds[30000][1])
        i2, d2 = c2.nn_search(clouds[12000][0], clouds[30000][1])
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
        assert same(c1.icp_register(*clouds[150000], p), want[150000])


def test_a_failed_load_leaves_the_context_usable(capi, monkeypatch):
    # mi_icp_load puts the fixed cloud's upload and index builds on auxiliary streams and joins them into the main stream at the end --
    # an early return in between (a refused hierarchy, a failed reserve) used to leave them un-joined.  mi_selftest_fail_loads(ctx, 2) (an explicit call: ADVICE r05) makes the context's
    # first two index builds fail behind the upload; every exit path now joins the lanes: the failed calls raise, the next one on the SAME context
    # registers the clouds bit for bit as a fresh context does, also at another size (buffers outgrown and retired in between).
    from conftest import synth_cloud
    b1, a1 = synth_cloud(60000, seed=3)[:2]
    b2, a2 = synth_cloud(150000, seed=4)[:2]
    p = capi.icp_params(eps=0.0, max_iterations=6)
    with capi.Context(0) as bad:
        bad.selftest_fail_loads(2)
        with capi.Context(0) as fresh:
            with pytest.raises(capi.MiSlamError):
                bad.icp_register(b1, a1, p)
            with pytest.raises(capi.MiSlamError):
                bad.icp_register(b2, a2, p)
            for b, a in ((b2, a2), (b1, a1)):
                x, y = bad.icp_register(b, a, p), fresh.icp_register(b, a, p)
                assert x[2] == y[2] == 6 and np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[3] == y[3]
