"""ctypes binding of libmislam.so -- the C ABI declared in include/mi_slam.h.

This module is plumbing for the Python-side callers in this repository (tests, bench.py, __graft_entry__): it passes host
numpy buffers straight through to the C entry points and adds nothing of its own.  There is deliberately no fallback: if
the shared library is missing, or no HIP device is usable, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MISLAM_LIB") or os.path.join(_HERE, "libmislam.so")   # MISLAM_LIB: developer override (variant builds)

MI_OK = 0
DIST_CPU_ROUNDING, DIST_FMA = 0, 1
COMPOSE_CPU_ADDITIVE, COMPOSE_EXACT = 0, 1
NN_AUTO, NN_BRUTEFORCE, NN_TREE, NN_GRID = 0, 1, 2, 3
SHARD_AUTO, SHARD_TARGET, SHARD_SOURCE = 0, 1, 2
SUM_EXACT, SUM_CPU_SEQUENTIAL = 0, 1
SIGMA2_EXACT, SIGMA2_CPU_SEQUENTIAL = 0, 1
ESTEP_DEFAULT, ESTEP_CPU_SEQUENTIAL = 0, 1
NN_INDEX_MIN_POINTS = 10000         # MI_NN_AUTO switches to the cell grid at this many fixed points (mi_slam.h MI_NN_INDEX_MIN_POINTS)
STOP_RUNNING, STOP_CONVERGED, STOP_MAX_ITERATIONS, STOP_NO_PAIRS, STOP_ERROR_INCREASED, STOP_TOLERANCE, STOP_SIGMA = range(7)
(KERNEL_NN, KERNEL_MOMENTS, KERNEL_SOLVE, KERNEL_TRANSFORM, KERNEL_FINALIZE, KERNEL_ALLREDUCE, KERNEL_CPD_DENOM,
 KERNEL_CPD_CONTRACT, KERNEL_CPD_MSTEP, KERNEL_CPD_FGT) = range(10)
KERNEL_NAMES = ["nn", "moments", "solve", "transform", "finalize", "allreduce", "cpd_denom", "cpd_contract", "cpd_mstep", "cpd_fgt"]
CPD_APPROX_NONE, CPD_APPROX_FULL, CPD_APPROX_HYBRID = 0, 1, 2
UNIQUE_ID_BYTES = 128
EXCHANGE_MIN_U64, EXCHANGE_SUM_F64 = 0, 1
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)      # mi_exchange_fn

# every symbol include/mi_slam.h declares (tests check that the library exports each of them)
EXPORTS = [
    "mi_abi_version", "mi_last_error", "mi_device_count", "mi_ctx_create", "mi_dist_unique_id", "mi_ctx_create_dist",
    "mi_ctx_create_exchange", "mi_runtime_info", "mi_ctx_preload", "mi_ctx_rank", "mi_dist_info", "mi_shard_range", "mi_source_share", "mi_pack_key", "mi_unpack_key", "mi_ctx_destroy", "mi_ctx_synchronize", "mi_icp_params_default", "mi_icp_params_cuda_slam",
    "mi_icp_register", "mi_icp_load", "mi_icp_reset", "mi_icp_run", "mi_icp_auto_batch", "mi_icp_result", "mi_nn_search", "mi_nn_search_ex", "mi_cross_moments", "mi_kabsch",
    "mi_transform_mse", "mi_cpd_params_default", "mi_cpd_register", "mi_cpd_sigma_squared", "mi_cpd_sigma_squared_mode", "mi_cpd_estep",
    "mi_cpd_estep_truncated", "mi_cpd_estep_fgt", "mi_fgt_kcenter", "mi_fgt_kcenter_guided", "mi_fgt_tables", "mi_nicp_params_default", "mi_nicp_register",
    "mi_prepare_params_default", "mi_prepare_cloud",
    "mi_cpd_mstep", "mi_profile_enable", "mi_profile_select", "mi_profile_reset", "mi_profile_get", "mi_icp_load_times", "mi_profile_search_stats", "mi_profile_search_phases", "mi_selftest_sort_pairs", "mi_selftest_fail_loads", "mi_nn_kernel_name",
]


class IcpParams(C.Structure):
    _fields_ = [("eps", C.c_float), ("max_iterations", C.c_int), ("max_distance_squared", C.c_float),
                ("dist_mode", C.c_int), ("compose_mode", C.c_int), ("filter_pairs", C.c_int),
                ("abort_on_increase", C.c_int), ("sync_every", C.c_int), ("verbose", C.c_int), ("nn_mode", C.c_int),
                ("shard_mode", C.c_int), ("sum_mode", C.c_int), ("reserved", C.c_int * 4)]


class CpdParams(C.Structure):
    _fields_ = [("eps", C.c_float), ("weight", C.c_float), ("const_scale", C.c_int), ("max_iterations", C.c_int),
                ("tolerance", C.c_float), ("sigma2_init", C.c_float), ("sync_every", C.c_int), ("verbose", C.c_int),
                ("approximation", C.c_int), ("fgt_ratio_of_far_field", C.c_float), ("fgt_order_of_truncation", C.c_int),
                ("sigma2_mode", C.c_int), ("estep_mode", C.c_int), ("reserved", C.c_int * 3)]


class NicpParams(C.Structure):
    _fields_ = [("eps", C.c_float), ("max_repetitions", C.c_int), ("approximation", C.c_int), ("verbose", C.c_int),
                ("reserved", C.c_int * 4)]


class PrepareParams(C.Structure):
    _fields_ = [("has_spread", C.c_int), ("spread", C.c_float), ("noise_intensity", C.c_float), ("has_transform", C.c_int),
                ("rotation", C.c_float * 9), ("translation", C.c_float * 3), ("reserved", C.c_int * 4)]


class MiSlamError(RuntimeError):
    pass


_lib = None
_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int)
_u8 = C.POINTER(C.c_ubyte)


def lib():
    """Load libmislam.so (raises if it has not been built -- run __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MiSlamError("libmislam.so not built at %s: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.mi_last_error.restype = C.c_char_p
        _lib.mi_ctx_destroy.restype = None
        _lib.mi_icp_params_default.restype = None
        _lib.mi_icp_params_cuda_slam.restype = None
        _lib.mi_cpd_params_default.restype = None
        _lib.mi_prepare_params_default.restype = None
        _lib.mi_pack_key.restype = C.c_ulonglong
        _lib.mi_pack_key.argtypes = [C.c_float, C.c_int]
        _lib.mi_unpack_key.restype = None
        _lib.mi_unpack_key.argtypes = [C.c_ulonglong, _f, _i]
    return _lib


def _check(rc):
    if rc != MI_OK:
        raise MiSlamError("libmislam error %d: %s" % (rc, lib().mi_last_error().decode()))


def _fp(a):
    return a.ctypes.data_as(_f)


def _cloud(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] != 3:
        raise ValueError("cloud must be [n,3] float32")
    return a


def device_count():
    c = C.c_int(0)
    _check(lib().mi_device_count(C.byref(c)))
    return c.value


def icp_params(cuda_slam=False, **kw):
    p = IcpParams()
    (lib().mi_icp_params_cuda_slam if cuda_slam else lib().mi_icp_params_default)(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def cpd_params(**kw):
    p = CpdParams()
    lib().mi_cpd_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def nicp_params(**kw):
    p = NicpParams()
    lib().mi_nicp_params_default.restype = None
    lib().mi_nicp_params_default(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def fgt_tables(order):
    """Host-only: exponents (a, b, c) per monomial in the reference's graded order, C_k and the Horner slot (mi_fgt_tables)."""
    pd = C.c_int(0)
    _check(lib().mi_fgt_tables(int(order), None, None, None, C.byref(pd)))
    mono = np.empty(pd.value, np.uint32)
    ck = np.empty(pd.value, np.float32)
    slot = np.empty(pd.value, np.int32)
    _check(lib().mi_fgt_tables(int(order), mono.ctypes.data_as(C.POINTER(C.c_uint)), _fp(ck), slot.ctypes.data_as(_i), C.byref(pd)))
    exps = np.stack([mono & 0xff, (mono >> 8) & 0xff, (mono >> 16) & 0xff], axis=1).astype(np.int32)
    return exps, ck, slot


def shard_range(m_total, rank, world):
    lo, hi = C.c_int(0), C.c_int(0)
    _check(lib().mi_shard_range(m_total, rank, world, C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def icp_auto_batch(n_moving_total, m_fixed_total, world, source_sharded, every_pair_search):
    """Iterations mi_icp_run enqueues between host checks with sync_every = 0: a function of global sizes only."""
    f = lib().mi_icp_auto_batch
    f.argtypes = [C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_int]
    return int(f(n_moving_total, m_fixed_total, world, 1 if source_sharded else 0, 1 if every_pair_search else 0))


def source_share(n_total, rank, world):
    """Moving points rank `rank` works on under SHARD_SOURCE."""
    cnt = C.c_int(0)
    _check(lib().mi_source_share(n_total, rank, world, C.byref(cnt)))
    return cnt.value


def pack_key(d2, index):
    return int(lib().mi_pack_key(float(d2), int(index)))


def unpack_key(key):
    d2, idx = C.c_float(0), C.c_int(0)
    lib().mi_unpack_key(C.c_ulonglong(key), C.byref(d2), C.byref(idx))
    return d2.value, idx.value


def dist_unique_id():
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    _check(lib().mi_dist_unique_id(buf))
    return buf.raw


def runtime_info():
    """{hip_path, rccl_path, hip_runtime_version, rccl_version}: the shared objects libmislam.so's HIP / RCCL calls bind to in this
    process (mi_runtime_info; no device touched)."""
    hip, rccl = C.create_string_buffer(1024), C.create_string_buffer(1024)
    hv, rv = C.c_int(0), C.c_int(0)
    _check(lib().mi_runtime_info(hip, rccl, 1024, C.byref(hv), C.byref(rv)))
    return {"hip_path": hip.value.decode(), "rccl_path": rccl.value.decode(), "hip_runtime_version": hv.value, "rccl_version": rv.value}


def _T_to_Rt(T):
    M = np.array(T, dtype=np.float32).reshape(4, 4).T   # column-major -> M[row, col]
    return M[:3, :3].copy(), M[:3, 3].copy()


class Context:
    """mi_ctx handle.  Context(device), Context(device, rank, world, unique_id) for the multi-GPU path over RCCL, or
    Context(device, rank, world, exchange=fn) for the same path over the caller's transport: fn(array, kind) combines the
    numpy array (uint64 for EXCHANGE_MIN_U64, float64 for EXCHANGE_SUM_F64) in place across the ranks."""

    def __init__(self, device=0, rank=None, world=None, unique_id=None, exchange=None):
        self._h = C.c_void_p()
        self._exchange_cb = None
        if world is None:
            _check(lib().mi_ctx_create(device, C.byref(self._h)))
        elif exchange is not None:
            def trampoline(_user, buf, count, kind):
                try:
                    ctype = C.c_uint64 if kind == EXCHANGE_MIN_U64 else C.c_double
                    exchange(np.ctypeslib.as_array(C.cast(buf, C.POINTER(ctype)), shape=(count,)), kind)
                    return 0
                except Exception:       # an exception must not unwind through the C frames
                    import traceback
                    traceback.print_exc()
                    return 1
            self._exchange_cb = EXCHANGE_FN(trampoline)      # kept alive with the context
            _check(lib().mi_ctx_create_exchange(device, rank, world, self._exchange_cb, None, C.byref(self._h)))
        else:
            _check(lib().mi_ctx_create_dist(device, rank, world, unique_id, C.byref(self._h)))

    def close(self):
        if self._h:
            lib().mi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().mi_ctx_synchronize(self._h))

    def preload(self):
        """Load all device code now instead of lazily inside the first calls (mi_ctx_preload)."""
        _check(lib().mi_ctx_preload(self._h))

    def rank_world(self):
        r, w = C.c_int(0), C.c_int(0)
        _check(lib().mi_ctx_rank(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def dist_info(self):
        """(nranks, rank) as the transport reports them and the bit mask of ranks seen in one all-reduce (collective call)."""
        n, r, seen = C.c_int(0), C.c_int(0), C.c_ulonglong(0)
        _check(lib().mi_dist_info(self._h, C.byref(n), C.byref(r), C.byref(seen)))
        return n.value, r.value, int(seen.value)

    # ---- ICP
    def icp_register(self, before, after, params):
        before, after = _cloud(before), _cloud(after)
        T = (C.c_float * 16)()
        it, err = C.c_int(0), C.c_float(0)
        _check(lib().mi_icp_register(self._h, _fp(before), before.shape[0], _fp(after), after.shape[0], C.byref(params), T,
                                     C.byref(it), C.byref(err)))
        R, t = _T_to_Rt(T)
        return R, t, it.value, err.value

    def icp_load(self, before, after, params):
        before, after = _cloud(before), _cloud(after)
        _check(lib().mi_icp_load(self._h, _fp(before), before.shape[0], _fp(after), after.shape[0], C.byref(params)))

    def icp_reset(self):
        _check(lib().mi_icp_reset(self._h))

    def icp_run(self, max_new_iterations):
        done = C.c_int(0)
        _check(lib().mi_icp_run(self._h, max_new_iterations, C.byref(done)))
        return done.value

    def icp_result(self):
        T = (C.c_float * 16)()
        it, err, why = C.c_int(0), C.c_float(0), C.c_int(0)
        _check(lib().mi_icp_result(self._h, T, C.byref(it), C.byref(err), C.byref(why)))
        R, t = _T_to_Rt(T)
        return R, t, it.value, err.value, why.value

    # ---- primitives
    def nn_search(self, src, tgt, dist_mode=DIST_CPU_ROUNDING, nn_mode=NN_AUTO):
        src, tgt = _cloud(src), _cloud(tgt)
        n = src.shape[0]
        idx = np.empty(n, np.int32)
        d2 = np.empty(n, np.float32)
        _check(lib().mi_nn_search_ex(self._h, _fp(src), n, _fp(tgt), tgt.shape[0], dist_mode, nn_mode,
                                     idx.ctypes.data_as(_i), _fp(d2)))
        return idx, d2

    def kabsch(self, src, tgt, idx, keep=None):
        src, tgt = _cloud(src), _cloud(tgt)
        idx = np.ascontiguousarray(idx, np.int32)
        kp = None
        if keep is not None:
            keep = np.ascontiguousarray(keep, np.uint8)
            kp = keep.ctypes.data_as(_u8)
        R9 = (C.c_float * 9)()
        t3 = (C.c_float * 3)()
        used = C.c_int(0)
        _check(lib().mi_kabsch(self._h, _fp(src), src.shape[0], _fp(tgt), tgt.shape[0], idx.ctypes.data_as(_i), kp, R9, t3,
                               C.byref(used)))
        return np.array(R9, np.float32).reshape(3, 3).T.copy(), np.array(t3, np.float32), used.value

    def cross_moments(self, src, tgt, idx, keep=None):
        """{pairs, sum src, sum tgt, sum tgt_r src_c} over the kept pairs (src[i], tgt[idx[i]]), fp64."""
        src, tgt = _cloud(src), _cloud(tgt)
        idx = np.ascontiguousarray(idx, np.int32)
        kp = None
        if keep is not None:
            keep = np.ascontiguousarray(keep, np.uint8)
            kp = keep.ctypes.data_as(_u8)
        out = (C.c_double * 16)()
        _check(lib().mi_cross_moments(self._h, _fp(src), src.shape[0], _fp(tgt), tgt.shape[0], idx.ctypes.data_as(_i), kp, out))
        return np.array(out, np.float64)

    def transform_mse(self, src, R, t, tgt=None, idx=None, keep=None, divide_by_pairs=True, want_cloud=True):
        src = _cloud(src)
        n = src.shape[0]
        R9 = np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)
        t3 = np.ascontiguousarray(t, np.float32)
        out = np.empty((n, 3), np.float32) if want_cloud else None
        mse = C.c_float(0)
        tp, m, ip, kp = None, 0, None, None
        if tgt is not None:
            tgt = _cloud(tgt)
            tp, m = _fp(tgt), tgt.shape[0]
            idx = np.ascontiguousarray(idx, np.int32)
            ip = idx.ctypes.data_as(_i)
            if keep is not None:
                keep = np.ascontiguousarray(keep, np.uint8)
                kp = keep.ctypes.data_as(_u8)
        _check(lib().mi_transform_mse(self._h, _fp(src), n, _fp(R9), _fp(t3), tp, m, ip, kp, 1 if divide_by_pairs else 0,
                                      _fp(out) if want_cloud else None, C.byref(mse) if tgt is not None else None))
        return out, (mse.value if tgt is not None else None)

    # ---- CPD
    def cpd_register(self, before, after, params):
        before, after = _cloud(before), _cloud(after)
        T = (C.c_float * 16)()
        it, err, sc = C.c_int(0), C.c_float(0), C.c_float(0)
        _check(lib().mi_cpd_register(self._h, _fp(before), before.shape[0], _fp(after), after.shape[0], C.byref(params), T,
                                     C.byref(sc), C.byref(it), C.byref(err)))
        sR, t = _T_to_Rt(T)
        return sR, t, sc.value, it.value, err.value

    def cpd_sigma_squared(self, before, after, mode=SIGMA2_EXACT):
        before, after = _cloud(before), _cloud(after)
        s = C.c_float(0)
        _check(lib().mi_cpd_sigma_squared_mode(self._h, _fp(before), before.shape[0], _fp(after), after.shape[0], mode, C.byref(s)))
        return s.value

    def cpd_estep(self, y, x, constant, sigma2):
        y, x = _cloud(y), _cloud(x)
        m, n = y.shape[0], x.shape[0]
        p1 = np.empty(m, np.float32)
        pt1 = np.empty(n, np.float32)
        px = np.empty((m, 3), np.float32)
        L = C.c_float(0)
        _check(lib().mi_cpd_estep(self._h, _fp(y), m, _fp(x), n, C.c_float(constant), C.c_float(sigma2), _fp(p1), _fp(pt1),
                                  _fp(px), C.byref(L)))
        return p1, pt1, px, L.value

    def cpd_estep_truncated(self, y, x, constant, sigma2, truncate=1e-3):
        y, x = _cloud(y), _cloud(x)
        m, n = y.shape[0], x.shape[0]
        p1, pt1, px, L = np.empty(m, np.float32), np.empty(n, np.float32), np.empty((m, 3), np.float32), C.c_float(0)
        _check(lib().mi_cpd_estep_truncated(self._h, _fp(y), m, _fp(x), n, C.c_float(constant), C.c_float(sigma2),
                                            C.c_float(truncate), _fp(p1), _fp(pt1), _fp(px), C.byref(L)))
        return p1, pt1, px, L.value

    def cpd_estep_fgt(self, y, x, weight, sigma2, sigma2_init, ratio_of_far_field=10.0, order_of_truncation=8):
        y, x = _cloud(y), _cloud(x)
        m, n = y.shape[0], x.shape[0]
        p1, pt1, px, L = np.empty(m, np.float32), np.empty(n, np.float32), np.empty((m, 3), np.float32), C.c_float(0)
        _check(lib().mi_cpd_estep_fgt(self._h, _fp(y), m, _fp(x), n, C.c_float(weight), C.c_float(sigma2), C.c_float(sigma2_init),
                                      C.c_float(ratio_of_far_field), int(order_of_truncation), _fp(p1), _fp(pt1), _fp(px),
                                      C.byref(L)))
        return p1, pt1, px, L.value

    def fgt_kcenter(self, cloud, K):
        cloud = _cloud(cloud)
        centers = np.empty((K, 3), np.float32)
        cluster = np.empty(cloud.shape[0], np.int32)
        _check(lib().mi_fgt_kcenter(self._h, _fp(cloud), cloud.shape[0], int(K), _fp(centers), cluster.ctypes.data_as(_i)))
        return centers, cluster

    def fgt_kcenter_guided(self, cloud, K, guess):
        """-> centers, cluster, picked (the sweep's choices), verified (leading entries of `guess` that were the sweep's own; -1: no replay)"""
        cloud = _cloud(cloud)
        guess = np.ascontiguousarray(guess, np.int32)
        centers = np.empty((K, 3), np.float32)
        cluster = np.empty(cloud.shape[0], np.int32)
        picked = np.empty(K, np.int32)
        verified = C.c_int(-2)
        _check(lib().mi_fgt_kcenter_guided(self._h, _fp(cloud), cloud.shape[0], int(K), guess.ctypes.data_as(_i), int(guess.shape[0]), _fp(centers),
                                           cluster.ctypes.data_as(_i), picked.ctypes.data_as(_i), C.byref(verified)))
        return centers, cluster, picked, verified.value

    def cpd_mstep(self, before, after, p1, pt1, px, const_scale, scale=1.0, sigma2=0.0):
        before, after = _cloud(before), _cloud(after)
        p1 = np.ascontiguousarray(p1, np.float32)
        pt1 = np.ascontiguousarray(pt1, np.float32)
        px = np.ascontiguousarray(px, np.float32)
        R9 = (C.c_float * 9)()
        t3 = (C.c_float * 3)()
        s, s2 = C.c_float(scale), C.c_float(sigma2)
        _check(lib().mi_cpd_mstep(self._h, _fp(before), before.shape[0], _fp(after), after.shape[0], _fp(p1), _fp(pt1), _fp(px),
                                  1 if const_scale else 0, R9, t3, C.byref(s), C.byref(s2)))
        return np.array(R9, np.float32).reshape(3, 3).T.copy(), np.array(t3, np.float32), s.value, s2.value

    # ---- NICP
    def nicp_register(self, before, after, params, order_heads, subcloud_idx=None):
        """order_heads: [repetitions, 3] int32 (first three entries of each repetition's permutation); subcloud_idx None = whole cloud."""
        before, after = _cloud(before), _cloud(after)
        heads = np.ascontiguousarray(order_heads, np.int32)
        reps = 20 if params.max_repetitions == -1 else params.max_repetitions
        if heads.shape != (reps, 3):
            raise ValueError("order_heads must be [%d, 3]" % reps)
        if subcloud_idx is not None:
            subcloud_idx = np.ascontiguousarray(subcloud_idx, np.int32)
        sn = before.shape[0] if subcloud_idx is None else len(subcloud_idx)
        T = (C.c_float * 16)()
        it, err = C.c_int(0), C.c_float(0)
        _check(lib().mi_nicp_register(self._h, _fp(before), before.shape[0], _fp(after), after.shape[0], C.byref(params),
                                      heads.ctypes.data_as(_i), subcloud_idx.ctypes.data_as(_i) if subcloud_idx is not None else None,
                                      sn, T, C.byref(it), C.byref(err)))
        R, t = _T_to_Rt(T)
        return R, t, it.value, err.value

    # ---- input stage
    def prepare_cloud(self, raw, subcloud_idx=None, shuffle_idx=None, noise_rows=None, noise_unit=None, noise_intensity=0.0,
                      outlier_unit=None, spread=None, R=None, t=None):
        """GetCloudsFromConfig's stages for one cloud (mi_prepare_cloud).  R (3x3, row = output component), t: the known
        transformation, or None.  Returns the prepared cloud [(size + outliers), 3]."""
        raw = _cloud(raw)
        opt_i = lambda a: None if a is None else np.ascontiguousarray(a, np.int32)
        opt_f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).reshape(-1)
        sub, shuf, rows, nu, ou = opt_i(subcloud_idx), opt_i(shuffle_idx), opt_i(noise_rows), opt_f(noise_unit), opt_f(outlier_unit)
        n = raw.shape[0] if sub is None else len(sub)
        if shuf is not None and len(shuf) != n:
            raise ValueError("shuffle_idx must have %d entries" % n)
        n_noise = 0 if rows is None else len(rows)
        if n_noise and (nu is None or len(nu) != 3 * n_noise):
            raise ValueError("noise_unit must be [%d, 3]" % n_noise)
        n_out = 0 if ou is None else len(ou) // 3
        p = PrepareParams()
        lib().mi_prepare_params_default(C.byref(p))
        if spread is not None:
            p.has_spread, p.spread = 1, spread
        p.noise_intensity = noise_intensity
        if R is not None:
            p.has_transform = 1
            p.rotation[:] = np.asarray(R, np.float32).T.reshape(9).tolist()        # column-major
            p.translation[:] = np.asarray(t, np.float32).tolist()
        out = np.empty((n + n_out, 3), np.float32)
        got = C.c_int(0)
        ip = lambda a: None if a is None else a.ctypes.data_as(_i)
        fp = lambda a: None if a is None else _fp(a)
        _check(lib().mi_prepare_cloud(self._h, _fp(raw), raw.shape[0], ip(sub), n, ip(shuf), ip(rows), fp(nu), n_noise, fp(ou), n_out,
                                      C.byref(p), _fp(out), C.byref(got)))
        assert got.value == n + n_out
        return out

    # ---- profiling
    def profile_enable(self, on=True):
        _check(lib().mi_profile_enable(self._h, 1 if on else 0))

    def search_stats(self, enable):
        """Counters of the cell-grid search since they were last enabled: (candidates, rows, to_hierarchy, points, nodes, leaves,
        walking waves, longest single walk)."""
        out = (C.c_ulonglong * 8)()
        _check(lib().mi_profile_search_stats(self._h, 1 if enable else 0, out))
        return tuple(int(v) for v in out)

    SEARCH_PHASES = ("waves", "scan_waves", "walk_only_waves", "block_batches", "block_dealt", "block_deal_passes", "block_deal_writes", "block_lockstep_trips",
                     "rest_rounds", "rest_dealt", "rest_deal_passes", "rest_deal_writes", "rest_lockstep_trips", "rest_batches4", "rest_waves",
                     "walk_leaf_hits", "walk_leaf_children", "walk_votes", "walk_pops", "walk_leaf_offers")

    def search_phases(self):
        """Loop trip counts of the counting build since search_stats(True), summed over waves (mi_profile_search_phases) -> dict."""
        out = (C.c_ulonglong * 20)()
        _check(lib().mi_profile_search_phases(self._h, out))
        return {k: int(out[i]) for i, k in enumerate(self.SEARCH_PHASES)}

    def selftest_fail_loads(self, n):
        """Test hook: the next n index builds of this context fail on purpose (mi_selftest_fail_loads)."""
        _check(lib().mi_selftest_fail_loads(self._h, int(n)))

    def selftest_sort_pairs(self, keys, values, bits=30):
        """The library's device radix sort on host arrays: (sorted keys, values carried along), stable."""
        k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
        v = np.ascontiguousarray(values, dtype=np.int32).copy()
        assert k.shape == v.shape and k.ndim == 1
        _check(lib().mi_selftest_sort_pairs(self._h, k.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p), int(k.size), int(bits)))
        return k, v

    def nn_kernel_name(self, n_moving, m_fixed_local, nn_mode=NN_AUTO):
        lib().mi_nn_kernel_name.restype = C.c_char_p
        return lib().mi_nn_kernel_name(self._h, n_moving, m_fixed_local, nn_mode).decode()

    def profile_select(self, kernels=None):
        """Restrict the event timing to these kernels (KERNEL_* ids); None = all."""
        mask = 0xffffffff if kernels is None else sum(1 << k for k in kernels)
        _check(lib().mi_profile_select(self._h, C.c_uint(mask)))

    def profile_reset(self):
        _check(lib().mi_profile_reset(self._h))

    def icp_load_times(self):
        """ms per stage of the last icp_load: workspace, moving upload, moving order, fixed upload, hierarchy, grid, reset, total."""
        out = (C.c_double * 8)()
        _check(lib().mi_icp_load_times(self._h, out))
        return dict(zip(("workspace", "upload_moving", "order_moving", "upload_fixed", "hierarchy", "grid", "reset", "total"), list(out)))

    def profile_get(self, kernel):
        ms, n = C.c_double(0), C.c_longlong(0)
        _check(lib().mi_profile_get(self._h, kernel, C.byref(ms), C.byref(n)))
        return ms.value, n.value
