"""TEST INFRASTRUCTURE ONLY.  ctypes binding of oracle/liboracle.so (the plain-C restatement, oracle/slam_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

DIST_CPU, DIST_FMA = 0, 1
COMPOSE_CPU_ADDITIVE, COMPOSE_EXACT = 0, 1

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)


class IcpParams(C.Structure):
    _fields_ = [("eps", C.c_float), ("max_distance_squared", C.c_float), ("max_iterations", C.c_int),
                ("threads", C.c_int), ("dist_mode", C.c_int), ("compose_mode", C.c_int),
                ("abort_on_increase", C.c_int), ("filter_pairs", C.c_int)]


class CpdParams(C.Structure):
    _fields_ = [("eps", C.c_float), ("weight", C.c_float), ("const_scale", C.c_int), ("max_iterations", C.c_int),
                ("tolerance", C.c_float)]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.oracle_mse_indexed.restype = C.c_float
        _lib.oracle_cpd_sigma_squared.restype = C.c_float
        _lib.oracle_cpd_constant.restype = C.c_float
        _lib.oracle_filter_pairs.restype = C.c_int
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f)


def _ip(a):
    return a.ctypes.data_as(_i)


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def _col9(R):
    return np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)


def _from_col9(r):
    return r.reshape(3, 3).T.copy()


def nn_search(before, after, threads=0, dist_mode=DIST_CPU):
    before, after = _cloud(before), _cloud(after)
    n, m = before.shape[0], after.shape[0]
    idx = np.empty(n, np.int32)
    d2 = np.empty(n, np.float32)
    lib().oracle_nn_search(_fp(before), n, _fp(after), m, threads, dist_mode, _ip(idx), _fp(d2))
    return idx, d2


def filter_pairs(d2, max_distance_squared):
    d2 = np.ascontiguousarray(d2, np.float32)
    ib = np.empty(len(d2), np.int32)
    k = lib().oracle_filter_pairs(_fp(d2), len(d2), C.c_float(max_distance_squared), _ip(ib))
    return ib[:k].copy()


def jacobi_svd3(A):
    a = np.ascontiguousarray(A, np.float32).reshape(9)
    u = np.empty(9, np.float32)
    s = np.empty(3, np.float32)
    v = np.empty(9, np.float32)
    lib().oracle_jacobi_svd3(_fp(a), _fp(u), _fp(s), _fp(v))
    return u.reshape(3, 3), s, v.reshape(3, 3)


def least_squares_svd(before_pts, after_pts):
    before_pts, after_pts = _cloud(before_pts), _cloud(after_pts)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    lib().oracle_least_squares_svd(_fp(before_pts), _fp(after_pts), before_pts.shape[0], _fp(r), _fp(t))
    return _from_col9(r), t


def transform_cloud(cloud, R, t, scale=None):
    cloud = _cloud(cloud)
    out = np.empty_like(cloud)
    r = _col9(R)
    tt = np.ascontiguousarray(t, np.float32)
    lib().oracle_transform_cloud(_fp(cloud), cloud.shape[0], _fp(r), _fp(tt), C.c_float(1.0 if scale is None else scale),
                                 0 if scale is None else 1, _fp(out))
    return out


def mse_indexed(before, after, ib, ia):
    before, after = _cloud(before), _cloud(after)
    ib = np.ascontiguousarray(ib, np.int32)
    ia = np.ascontiguousarray(ia, np.int32)
    return float(lib().oracle_mse_indexed(_fp(before), _fp(after), _ip(ib), _ip(ia), len(ib)))


def icp(before, after, eps=1e-3, max_distance_squared=1000.0, max_iterations=-1, threads=0, dist_mode=DIST_CPU,
        compose_mode=COMPOSE_CPU_ADDITIVE, abort_on_increase=False, filter_pairs=True, trace_cap=0):
    before, after = _cloud(before), _cloud(after)
    p = IcpParams(eps, max_distance_squared, max_iterations, threads, dist_mode, compose_mode,
                  1 if abort_on_increase else 0, 1 if filter_pairs else 0)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    it = C.c_int(0)
    err = C.c_float(0)
    trace = np.zeros((max(trace_cap, 1), 14), np.float32)
    tl = C.c_int(0)
    lib().oracle_icp(_fp(before), before.shape[0], _fp(after), after.shape[0], C.byref(p), _fp(r), _fp(t),
                     C.byref(it), C.byref(err), _fp(trace) if trace_cap else None, trace_cap, C.byref(tl))
    out = (_from_col9(r), t, it.value, err.value)
    if trace_cap:
        return out + (trace[:min(tl.value, trace_cap)].copy(),)
    return out


def cpd_sigma_squared(before, after):
    before, after = _cloud(before), _cloud(after)
    return float(lib().oracle_cpd_sigma_squared(_fp(before), before.shape[0], _fp(after), after.shape[0]))


def cpd_constant(sigma_squared, weight, m_before, n_after):
    return float(lib().oracle_cpd_constant(C.c_float(sigma_squared), C.c_float(weight), m_before, n_after))


def cpd_estep(transformed, after, constant, sigma_squared):
    transformed, after = _cloud(transformed), _cloud(after)
    m, n = transformed.shape[0], after.shape[0]
    p1 = np.empty(m, np.float32)
    pt1 = np.empty(n, np.float32)
    px = np.empty((m, 3), np.float32)
    L = C.c_float(0)
    lib().oracle_cpd_estep(_fp(transformed), m, _fp(after), n, C.c_float(constant), C.c_float(sigma_squared),
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
    lib().oracle_cpd_mstep(_fp(before), before.shape[0], _fp(after), after.shape[0], _fp(p1), _fp(pt1), _fp(px),
                           1 if const_scale else 0, _fp(r), _fp(t), C.byref(s), C.byref(s2))
    return _from_col9(r), t, s.value, s2.value


def cpd(before, after, eps=1e-3, weight=0.3, const_scale=False, max_iterations=50, tolerance=1e-3, trace_cap=0):
    before, after = _cloud(before), _cloud(after)
    p = CpdParams(eps, weight, 1 if const_scale else 0, max_iterations, tolerance)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    it = C.c_int(0)
    err = C.c_float(0)
    trace = np.zeros((max(trace_cap, 1), 16), np.float32)
    tl = C.c_int(0)
    lib().oracle_cpd(_fp(before), before.shape[0], _fp(after), after.shape[0], C.byref(p), _fp(r), _fp(t),
                     C.byref(it), C.byref(err), _fp(trace) if trace_cap else None, trace_cap, C.byref(tl))
    out = (_from_col9(r), t, it.value, err.value)
    if trace_cap:
        return out + (trace[:min(tl.value, trace_cap)].copy(),)
    return out


# ---- non-iterative registration (oracle/nicp_oracle.c) ----
def nicp_single(before, after):
    before, after = _cloud(before), _cloud(after)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    e = C.c_float(0)
    lib().oracle_nicp_single(_fp(before), before.shape[0], _fp(after), after.shape[0], _fp(r), _fp(t), C.byref(e))
    return _from_col9(r), t, e.value


def nicp(before, after, perms, subcloud_idx=None, eps=1e-3, max_repetitions=20, approximation=0):
    """perms: [max_repetitions, min(m, n)] int32, one permutation per repetition; subcloud_idx None = the whole cloud."""
    before, after = _cloud(before), _cloud(after)
    perms = np.ascontiguousarray(perms, np.int32)
    assert perms.shape == (max_repetitions, min(len(before), len(after)))
    if subcloud_idx is not None:
        subcloud_idx = np.ascontiguousarray(subcloud_idx, np.int32)
    sn = len(before) if subcloud_idx is None else len(subcloud_idx)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    reps = C.c_int(0)
    err = C.c_float(0)
    lib().oracle_nicp(_fp(before), before.shape[0], _fp(after), after.shape[0], C.c_float(eps), max_repetitions, approximation,
                      _ip(subcloud_idx) if subcloud_idx is not None else None, sn, _ip(perms), _fp(r), _fp(t), C.byref(reps),
                      C.byref(err))
    return _from_col9(r), t, reps.value, err.value


# ---- Fast Gauss Transform E-step and the full / hybrid drivers (oracle/fgt_oracle.c) ----
APPROX_NONE, APPROX_FULL, APPROX_HYBRID = 0, 1, 2


def fgt_pd(p):
    return int(lib().oracle_fgt_pd(int(p)))


def fgt_ck(p):
    ck = np.empty(fgt_pd(p), np.float32)
    lib().oracle_fgt_ck(int(p), _fp(ck))
    return ck


def fgt_kcenter(cloud, K):
    cloud = _cloud(cloud)
    xc = np.empty((K, 3), np.float32)
    indx = np.empty(cloud.shape[0], np.int32)
    lib().oracle_fgt_kcenter(_fp(cloud), cloud.shape[0], K, _fp(xc), _ip(indx))
    return xc, indx


def fgt_model(cloud, weights, sigma, K, p):
    cloud = _cloud(cloud)
    weights = np.ascontiguousarray(weights, np.float32)
    xc = np.empty((K, 3), np.float32)
    ak = np.empty((K, fgt_pd(p)), np.float32)
    lib().oracle_fgt_model(_fp(cloud), cloud.shape[0], _fp(weights), C.c_float(sigma), K, p, _fp(xc), _fp(ak))
    return xc, ak


def fgt_predict(cloud, xc, ak, sigma, e_param, p):
    cloud = _cloud(cloud)
    xc = np.ascontiguousarray(xc, np.float32)
    ak = np.ascontiguousarray(ak, np.float32)
    v = np.empty(cloud.shape[0], np.float32)
    lib().oracle_fgt_predict(_fp(cloud), cloud.shape[0], _fp(xc), _fp(ak), C.c_float(sigma), C.c_float(e_param),
                             ak.shape[0], p, _fp(v))
    return v


def cpd_fgt_clusters(m, n, sigma_squared, sigma_squared_init):
    return int(lib().oracle_cpd_fgt_clusters(m, n, C.c_float(sigma_squared), C.c_float(sigma_squared_init)))


def cpd_fgt_ndi(sigma_squared, weight, m, n):
    lib().oracle_cpd_fgt_ndi.restype = C.c_float
    return float(lib().oracle_cpd_fgt_ndi(C.c_float(sigma_squared), C.c_float(weight), m, n))


def _estep_out(m, n):
    return np.empty(m, np.float32), np.empty(n, np.float32), np.empty((m, 3), np.float32), C.c_float(0)


def cpd_estep_fgt(transformed, after, weight, sigma_squared, sigma_squared_init, ratio_of_far_field=10.0,
                  order_of_truncation=8.0):
    transformed, after = _cloud(transformed), _cloud(after)
    m, n = transformed.shape[0], after.shape[0]
    p1, pt1, px, L = _estep_out(m, n)
    lib().oracle_cpd_estep_fgt(_fp(transformed), m, _fp(after), n, C.c_float(weight), C.c_float(sigma_squared),
                               C.c_float(sigma_squared_init), C.c_float(ratio_of_far_field), C.c_float(order_of_truncation),
                               _fp(p1), _fp(pt1), _fp(px), C.byref(L))
    return p1, pt1, px, L.value


def cpd_estep_truncated(transformed, after, constant, sigma_squared, truncate=1e-3):
    transformed, after = _cloud(transformed), _cloud(after)
    m, n = transformed.shape[0], after.shape[0]
    p1, pt1, px, L = _estep_out(m, n)
    lib().oracle_cpd_estep_truncated(_fp(transformed), m, _fp(after), n, C.c_float(constant), C.c_float(sigma_squared),
                                     C.c_float(truncate), _fp(p1), _fp(pt1), _fp(px), C.byref(L))
    return p1, pt1, px, L.value


def cpd_approx(before, after, approximation, eps=1e-3, weight=0.3, const_scale=False, max_iterations=50, tolerance=1e-3,
               ratio_of_far_field=10.0, order_of_truncation=8.0, trace_cap=0):
    before, after = _cloud(before), _cloud(after)
    p = CpdParams(eps, weight, 1 if const_scale else 0, max_iterations, tolerance)
    r = np.empty(9, np.float32)
    t = np.empty(3, np.float32)
    it = C.c_int(0)
    err = C.c_float(0)
    trace = np.zeros((max(trace_cap, 1), 17), np.float32)
    tl = C.c_int(0)
    lib().oracle_cpd_approx(_fp(before), before.shape[0], _fp(after), after.shape[0], C.byref(p), approximation,
                            C.c_float(ratio_of_far_field), C.c_float(order_of_truncation), _fp(r), _fp(t),
                            C.byref(it), C.byref(err), _fp(trace) if trace_cap else None, trace_cap, C.byref(tl))
    out = (_from_col9(r), t, it.value, err.value)
    if trace_cap:
        return out + (trace[:min(tl.value, trace_cap)].copy(),)
    return out


# ---- input stage, Common::GetCloudsFromConfig for one cloud (oracle/prep_oracle.c) ----
def prepare_cloud(raw, subcloud_idx=None, shuffle_idx=None, noise_rows=None, noise_unit=None, noise_intensity=0.0, outlier_unit=None,
                  spread=None, R=None, t=None):
    """R (3x3, row = output component) and t: the known transformation, or None.  Returns the prepared cloud."""
    raw = _cloud(raw)
    opt_i = lambda a: None if a is None else np.ascontiguousarray(a, np.int32)
    opt_f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).reshape(-1)
    sub, shuf, rows, nu, ou = opt_i(subcloud_idx), opt_i(shuffle_idx), opt_i(noise_rows), opt_f(noise_unit), opt_f(outlier_unit)
    n = len(raw) if sub is None else len(sub)
    n_noise = 0 if rows is None else len(rows)
    n_out = 0 if ou is None else len(ou) // 3
    out = np.empty((n + n_out, 3), np.float32)
    rot = None if R is None else np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)      # column-major
    tr = None if R is None else np.ascontiguousarray(t, np.float32)
    ptr_i = lambda a: None if a is None else _ip(a)
    ptr_f = lambda a: None if a is None else _fp(a)
    got = lib().oracle_prepare_cloud(_fp(raw), len(raw), ptr_i(sub), n, ptr_i(shuf), ptr_i(rows), ptr_f(nu), n_noise,
                                     C.c_float(noise_intensity), ptr_f(ou), n_out, 0 if spread is None else 1,
                                     C.c_float(spread or 0.0), ptr_f(rot), ptr_f(tr), _fp(out))
    assert got == n + n_out
    return out
