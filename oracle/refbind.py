"""TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/_ref/libref_cpuslam.so -- the reference's own cpu-slam code
compiled by oracle/Makefile (see oracle/ref_shim.cpp for the reference function behind each call).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_cpuslam.so")

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)


def available():
    return os.path.exists(LIB_PATH)


_lib = None


def _flush():
    """The reference prints its per-iteration lines through C stdio: flush them while the caller's output capture is still active."""
    C.CDLL(None).fflush(None)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.ref_mse_indexed.restype = C.c_float
        _lib.ref_cpd_sigma_squared.restype = C.c_float
        _lib.ref_corresponding_points.restype = C.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def clouds_from_config(raw, spread, seed, rot9_rowmajor, trans3, scale=1.0):
    raw = _cloud(raw)
    n = raw.shape[0]
    before = np.empty((n, 3), np.float32)
    after = np.empty((n, 3), np.float32)
    rot = np.ascontiguousarray(rot9_rowmajor, dtype=np.float32)
    tr = np.ascontiguousarray(trans3, dtype=np.float32)
    lib().ref_clouds_from_config(_fp(raw), n, 0 if spread is None else 1, C.c_float(spread or 0.0), C.c_uint(seed),
                                 _fp(rot), _fp(tr), C.c_float(scale), _fp(before), _fp(after))
    return before, after


def corresponding_points(before, after, max_distance_squared, parallel=True):
    before, after = _cloud(before), _cloud(after)
    n, m = before.shape[0], after.shape[0]
    ib = np.empty(n, np.int32)
    ia = np.empty(n, np.int32)
    k = lib().ref_corresponding_points(_fp(before), n, _fp(after), m, C.c_float(max_distance_squared),
                                       1 if parallel else 0, _ip(ib), _ip(ia))
    return ib[:k].copy(), ia[:k].copy()


def least_squares_svd(before, after):
    before, after = _cloud(before), _cloud(after)
    assert before.shape == after.shape
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    lib().ref_least_squares_svd(_fp(before), _fp(after), before.shape[0], _fp(r), _fp(t))
    return r.reshape(3, 3).T.copy(), t  # column-major -> R[row, col]


def transform_cloud(cloud, R, t):
    cloud = _cloud(cloud)
    r = np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)  # to column-major
    tt = np.ascontiguousarray(t, dtype=np.float32)
    out = np.empty_like(cloud)
    lib().ref_transform_cloud(_fp(cloud), cloud.shape[0], _fp(r), _fp(tt), _fp(out))
    return out


def mse_indexed(before, after, ib, ia):
    before, after = _cloud(before), _cloud(after)
    ib = np.ascontiguousarray(ib, np.int32)
    ia = np.ascontiguousarray(ia, np.int32)
    return float(lib().ref_mse_indexed(_fp(before), before.shape[0], _fp(after), after.shape[0], _ip(ib), _ip(ia), len(ib)))


def icp(before, after, eps=1e-3, max_distance_squared=1000.0, max_iterations=-1, parallel=True):
    before, after = _cloud(before), _cloud(after)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    it = C.c_int(0)
    err = C.c_float(0)
    lib().ref_icp(_fp(before), before.shape[0], _fp(after), after.shape[0], C.c_float(eps),
                  C.c_float(max_distance_squared), max_iterations, 1 if parallel else 0, _fp(r), _fp(t),
                  C.byref(it), C.byref(err))
    _flush()
    return r.reshape(3, 3).T.copy(), t, it.value, err.value


def cpd_sigma_squared(before, after):
    before, after = _cloud(before), _cloud(after)
    return float(lib().ref_cpd_sigma_squared(_fp(before), before.shape[0], _fp(after), after.shape[0]))


def cpd_estep(transformed, after, constant, sigma_squared):
    transformed, after = _cloud(transformed), _cloud(after)
    m, n = transformed.shape[0], after.shape[0]
    p1 = np.empty(m, np.float32)
    pt1 = np.empty(n, np.float32)
    px = np.empty((m, 3), np.float32)
    L = C.c_float(0)
    lib().ref_cpd_estep(_fp(transformed), m, _fp(after), n, C.c_float(constant), C.c_float(sigma_squared),
                        _fp(p1), _fp(pt1), _fp(px), C.byref(L))
    return p1, pt1, px, L.value


def cpd_mstep(before, after, p1, pt1, px, const_scale, scale=1.0, sigma_squared=0.0):
    before, after = _cloud(before), _cloud(after)
    p1 = np.ascontiguousarray(p1, np.float32)
    pt1 = np.ascontiguousarray(pt1, np.float32)
    px = np.ascontiguousarray(px, np.float32)
    r = np.eye(3, dtype=np.float32).reshape(9).copy()
    t = np.zeros(3, np.float32)
    s = C.c_float(scale)
    s2 = C.c_float(sigma_squared)
    lib().ref_cpd_mstep(_fp(before), before.shape[0], _fp(after), after.shape[0], _fp(p1), _fp(pt1), _fp(px),
                        1 if const_scale else 0, _fp(r), _fp(t), C.byref(s), C.byref(s2))
    return r.reshape(3, 3).T.copy(), t, s.value, s2.value


def cpd(before, after, eps=1e-3, weight=0.3, const_scale=False, max_iterations=50, tolerance=1e-3, fgt=0,
        ratio_of_far_field=10.0, order_of_truncation=8.0):
    before, after = _cloud(before), _cloud(after)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    it = C.c_int(0)
    err = C.c_float(0)
    lib().ref_cpd(_fp(before), before.shape[0], _fp(after), after.shape[0], C.c_float(eps), C.c_float(weight),
                  1 if const_scale else 0, max_iterations, C.c_float(tolerance), fgt, C.c_float(ratio_of_far_field),
                  C.c_float(order_of_truncation), _fp(r), _fp(t), C.byref(it), C.byref(err))
    _flush()
    return r.reshape(3, 3).T.copy(), t, it.value, err.value


def cpd_estep_truncated(transformed, after, constant, sigma_squared, truncate=1e-3):
    transformed, after = _cloud(transformed), _cloud(after)
    m, n = transformed.shape[0], after.shape[0]
    p1 = np.empty(m, np.float32)
    pt1 = np.empty(n, np.float32)
    px = np.empty((m, 3), np.float32)
    L = C.c_float(0)
    lib().ref_cpd_estep_truncated(_fp(transformed), m, _fp(after), n, C.c_float(constant), C.c_float(sigma_squared),
                                  C.c_float(truncate), _fp(p1), _fp(pt1), _fp(px), C.byref(L))
    return p1, pt1, px, L.value


def cpd_estep_fgt(transformed, after, weight, sigma_squared, sigma_squared_init, ratio_of_far_field=10.0,
                  order_of_truncation=8.0):
    transformed, after = _cloud(transformed), _cloud(after)
    m, n = transformed.shape[0], after.shape[0]
    p1 = np.empty(m, np.float32)
    pt1 = np.empty(n, np.float32)
    px = np.empty((m, 3), np.float32)
    L = C.c_float(0)
    lib().ref_cpd_estep_fgt(_fp(transformed), m, _fp(after), n, C.c_float(weight), C.c_float(sigma_squared),
                            C.c_float(sigma_squared_init), C.c_float(ratio_of_far_field), C.c_float(order_of_truncation),
                            _fp(p1), _fp(pt1), _fp(px), C.byref(L))
    return p1, pt1, px, L.value


def fgt_pd(p):
    """nchoosek(p + 2, 3): number of monomials of total degree < p in 3 variables (fgt.cpp:69)."""
    return (p + 2) * (p + 1) * p // 6


def fgt_model(cloud, weights, sigma, K, p):
    cloud = _cloud(cloud)
    weights = np.ascontiguousarray(weights, np.float32)
    xc = np.empty((K, 3), np.float32)
    ak = np.empty(fgt_pd(p) * K, np.float32)
    lib().ref_fgt_model.restype = C.c_int
    pd = lib().ref_fgt_model(_fp(cloud), cloud.shape[0], _fp(weights), C.c_float(sigma), K, p, _fp(xc), _fp(ak))
    assert pd == fgt_pd(p)
    return xc, ak.reshape(K, pd)      # row k = column k of the reference's (pd x K) matrix


def fgt_predict(cloud, xc, ak, sigma, e_param, p):
    cloud = _cloud(cloud)
    xc = np.ascontiguousarray(xc, np.float32)
    ak = np.ascontiguousarray(ak, np.float32)
    K, pd = ak.shape
    v = np.empty(cloud.shape[0], np.float32)
    lib().ref_fgt_predict(_fp(cloud), cloud.shape[0], _fp(xc), _fp(ak), pd, C.c_float(sigma), C.c_float(e_param), K, p, _fp(v))
    return v


def nicp_single(before, after):
    before, after = _cloud(before), _cloud(after)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    e = C.c_float(0)
    lib().ref_nicp_single(_fp(before), before.shape[0], _fp(after), after.shape[0], _fp(r), _fp(t), C.byref(e))
    _flush()
    return r.reshape(3, 3).T.copy(), t, e.value


def nicp(before, after, eps=1e-3, max_repetitions=20, approximation=0, parallel=False, subcloud_size=1000, seed=666):
    before, after = _cloud(before), _cloud(after)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    reps = C.c_int(0)
    err = C.c_float(0)
    lib().ref_nicp(_fp(before), before.shape[0], _fp(after), after.shape[0], C.c_float(eps), max_repetitions, approximation,
                   1 if parallel else 0, subcloud_size, C.c_uint(seed), _fp(r), _fp(t), C.byref(reps), C.byref(err))
    _flush()
    return r.reshape(3, 3).T.copy(), t, reps.value, err.value


def nicp_continue(before, after, eps=1e-3, max_repetitions=20, approximation=0, parallel=False, subcloud_size=1000):
    """cpu-slam's NICP on the generator as the last clouds_from_config_* call of this process left it (what the reference's program does)."""
    before, after = _cloud(before), _cloud(after)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    reps = C.c_int(0)
    err = C.c_float(0)
    lib().ref_nicp_continue(_fp(before), before.shape[0], _fp(after), after.shape[0], C.c_float(eps), max_repetitions, approximation,
                            1 if parallel else 0, subcloud_size, _fp(r), _fp(t), C.byref(reps), C.byref(err))
    _flush()
    return r.reshape(3, 3).T.copy(), t, reps.value, err.value


def random_permutation(seed, size, skip=0):
    out = np.empty(size, np.int32)
    lib().ref_random_permutation(C.c_uint(seed), size, skip, out.ctypes.data_as(C.POINTER(C.c_int)))
    return out


# ---- the input stage: GetCloudsFromConfig after LoadCloud, and the random outcomes it draws ----
def clouds_from_config_full(raw_before, raw_after, seed, R, t, resize_before=None, resize_after=None, spread=None,
                            noise_before=None, noise_after=None, outliers_before=0, outliers_after=0):
    """noise_*: (affected share, intensity) or None.  R (3x3, row = output component), t: the known transformation of `after`.
    raw_after None = the same file for both clouds (common.cpp:141-142)."""
    raw_before = _cloud(raw_before)
    nbr = len(raw_before)
    if raw_after is not None:
        raw_after = _cloud(raw_after)
    nar = nbr if raw_after is None else len(raw_after)
    nb = (min(resize_before, nbr) if resize_before is not None else nbr) + outliers_before
    na = (min(resize_after, nar) if resize_after is not None else nar) + outliers_after
    before = np.empty((nb, 3), np.float32)
    after = np.empty((na, 3), np.float32)
    rot = np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)
    tr = np.ascontiguousarray(t, np.float32)
    cb, ca = C.c_int(0), C.c_int(0)
    lib().ref_clouds_from_config_full(
        _fp(raw_before), nbr, None if raw_after is None else _fp(raw_after), nar,
        -1 if resize_before is None else resize_before, -1 if resize_after is None else resize_after,
        0 if spread is None else 1, C.c_float(spread or 0.0), C.c_uint(seed),
        C.c_float(-1.0 if noise_before is None else noise_before[0]), C.c_float(0.0 if noise_before is None else noise_before[1]),
        C.c_float(-1.0 if noise_after is None else noise_after[0]), C.c_float(0.0 if noise_after is None else noise_after[1]),
        outliers_before, outliers_after, _fp(rot), _fp(tr), _fp(before), C.byref(cb), _fp(after), C.byref(ca))
    assert (cb.value, ca.value) == (nb, na)
    return before, after


def clouds_from_config_random(raw_before, raw_after, seed, rot_range, trans_range, resize_before=None, resize_after=None, spread=None):
    """GetCloudsFromConfig with configuration.TransformationParameters (the reference's benchmark sets): returns (before, after, R, t) with the
    random rotation (3x3, row = output component) and translation the reference drew."""
    raw_before = _cloud(raw_before)
    nbr = len(raw_before)
    if raw_after is not None:
        raw_after = _cloud(raw_after)
    nar = nbr if raw_after is None else len(raw_after)
    nb = min(resize_before, nbr) if resize_before is not None else nbr
    na = min(resize_after, nar) if resize_after is not None else nar
    before = np.empty((nb, 3), np.float32)
    after = np.empty((na, 3), np.float32)
    rot = np.empty(9, np.float32)
    tr = np.empty(3, np.float32)
    cb, ca = C.c_int(0), C.c_int(0)
    lib().ref_clouds_from_config_random(
        _fp(raw_before), nbr, None if raw_after is None else _fp(raw_after), nar,
        -1 if resize_before is None else resize_before, -1 if resize_after is None else resize_after,
        0 if spread is None else 1, C.c_float(spread or 0.0), C.c_uint(seed), C.c_float(rot_range), C.c_float(trans_range),
        _fp(before), C.byref(cb), _fp(after), C.byref(ca), _fp(rot), _fp(tr))
    assert (cb.value, ca.value) == (nb, na)
    return before, after, rot.reshape(3, 3).T.copy(), tr


def permutation_sequence(seed, sizes):
    sizes = np.ascontiguousarray(sizes, np.int32)
    out = np.empty(int(sizes.sum()), np.int32)
    lib().ref_permutation_sequence(C.c_uint(seed), _ip(sizes), len(sizes), _ip(out))
    return np.split(out, np.cumsum(sizes)[:-1])


def config_draws(n_before_raw, n_after_raw, seed, resize_before=None, resize_after=None, noise_before=None, noise_after=None,
                 outliers_before=0, outliers_after=0):
    """The random outcomes GetCloudsFromConfig (common.cpp:134-190) draws for these options, in the form mi_prepare_cloud and
    oracle_prepare_cloud take them: two dicts (before, after) with subcloud_idx, shuffle_idx, noise_rows, noise_unit,
    outlier_unit.  Index vectors: the reference's own generator (ref_permutation_sequence), consumed in the reference's order;
    unit draws: this process's C library rand() seeded like mainwrapper.cpp:17-18 -- the same one the reference build calls."""
    sub_b = resize_before is not None and resize_before < n_before_raw        # GetSubcloud draws only then (common.cpp:27-28)
    sub_a = resize_after is not None and resize_after < n_after_raw
    nb = resize_before if sub_b else n_before_raw
    na = resize_after if sub_a else n_after_raw
    sizes = ([n_before_raw] if sub_b else []) + ([n_after_raw] if sub_a else []) + [nb, na]
    sizes += ([nb] if noise_before is not None else []) + ([na] if noise_after is not None else [])
    perms = list(permutation_sequence(seed, sizes))
    b = {"subcloud_idx": perms.pop(0)[:nb].copy() if sub_b else None}
    a = {"subcloud_idx": perms.pop(0)[:na].copy() if sub_a else None}
    b["shuffle_idx"], a["shuffle_idx"] = perms.pop(0), perms.pop(0)

    def rows(n, noise):
        if noise is None:
            return None
        affected = int(np.clip(np.floor(np.float32(noise[0]) * np.float32(n) + np.float32(0.5)), 0, n))   # std::round, clamped
        return np.nonzero(perms.pop(0) < affected)[0].astype(np.int32)      # ApplyPermutation: flag[perm[i]]
    b["noise_rows"], a["noise_rows"] = rows(nb, noise_before), rows(na, noise_after)

    libc = C.CDLL("libc.so.6")
    libc.srand(C.c_uint(seed))

    def unit(count):        # static_cast<float>(rand()) / RAND_MAX   (testutils.cpp:10; RAND_MAX converts to 2^31 in fp32)
        return (np.array([libc.rand() for _ in range(3 * count)], np.int64).astype(np.float32) / np.float32(2147483648.0)).reshape(count, 3)
    for d in (b, a):
        d["noise_unit"] = None if d["noise_rows"] is None else unit(len(d["noise_rows"]))
    b["outlier_unit"] = unit(outliers_before) if outliers_before else None
    a["outlier_unit"] = unit(outliers_after) if outliers_after else None
    return b, a
