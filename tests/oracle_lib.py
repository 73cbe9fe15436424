"""ctypes access to the CPU oracle (oracle/liblssvm_oracle.so) and, where it was built, to the reference's own OpenMP
kernels (oracle/_ref/liblssvm_ref.so).  TEST INFRASTRUCTURE: imported only by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() -- never by the plssvm_amd package.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liblssvm_oracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "liblssvm_ref.so")
REF_RELEASE_SO = os.path.join(ROOT, "oracle", "_ref", "liblssvm_ref_release.so")  # the reference's Release flags: timing baseline only

KERNELS = {"linear": 0, "polynomial": 1, "rbf": 2}


class CgInfo(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("delta", C.c_double), ("delta0", C.c_double), ("target", C.c_double),
                ("avg_iter_ms", C.c_double), ("total_ms", C.c_double)]


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32", C.c_float
    if dtype == np.float64:
        return "f64", C.c_double
    raise TypeError(f"unsupported dtype {dtype}")


def _ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class CpuPath:
    """One CPU implementation of the path: prefix 'oracle' (restatement) or 'ref' (reference TUs)."""

    def __init__(self, so_path: str, prefix: str):
        self.lib = C.CDLL(so_path)
        self.prefix = prefix

    def _fn(self, name, suf):
        return getattr(self.lib, f"{self.prefix}_{name}_{suf}")

    @staticmethod
    def _kt(kernel):
        return KERNELS[kernel] if isinstance(kernel, str) else int(kernel)

    def kernel_function(self, kernel, xi, xj, degree=3, gamma=1.0, coef0=0.0):
        xi = np.ascontiguousarray(xi)
        xj = np.ascontiguousarray(xj, dtype=xi.dtype)
        suf, ct = _suffix(xi.dtype)
        fn = self._fn("kernel_function", suf)
        fn.restype = ct
        fn.argtypes = [C.c_int, C.c_int, ct, ct, C.POINTER(ct), C.POINTER(ct), C.c_size_t]
        return fn(self._kt(kernel), degree, gamma, coef0, _ptr(xi, ct), _ptr(xj, ct), xi.size)

    def q(self, kernel, X, degree=3, gamma=1.0, coef0=0.0):
        X = np.ascontiguousarray(X)
        suf, ct = _suffix(X.dtype)
        N, d = X.shape
        q = np.zeros(N - 1, dtype=X.dtype)
        fn = self._fn("q", suf)
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int, ct, ct, C.POINTER(ct), C.c_size_t, C.c_size_t, C.POINTER(ct)]
        fn(self._kt(kernel), degree, gamma, coef0, _ptr(X, ct), N, d, _ptr(q, ct))
        return q

    def matvec(self, kernel, X, q, dvec, ret, QA_cost, cost, add, degree=3, gamma=1.0, coef0=0.0):
        """ret += add * Abar * dvec; `cost` is already 1/C.  Returns the updated copy of ret."""
        X = np.ascontiguousarray(X)
        suf, ct = _suffix(X.dtype)
        N, d = X.shape
        q = np.ascontiguousarray(q, dtype=X.dtype)
        dvec = np.ascontiguousarray(dvec, dtype=X.dtype)
        out = np.array(ret, dtype=X.dtype, copy=True)
        fn = self._fn("matvec", suf)
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int, ct, ct, C.POINTER(ct), C.c_size_t, C.c_size_t, C.POINTER(ct), C.POINTER(ct), C.POINTER(ct), ct, ct, ct]
        fn(self._kt(kernel), degree, gamma, coef0, _ptr(X, ct), N, d, _ptr(q, ct), _ptr(dvec, ct), _ptr(out, ct), QA_cost, cost, add)
        return out

    def matvec_rows(self, kernel, X, q, dvec, ret, QA_cost, cost, add, row_begin, row_end, degree=3, gamma=1.0, coef0=0.0):
        X = np.ascontiguousarray(X)
        suf, ct = _suffix(X.dtype)
        N, d = X.shape
        q = np.ascontiguousarray(q, dtype=X.dtype)
        dvec = np.ascontiguousarray(dvec, dtype=X.dtype)
        out = np.array(ret, dtype=X.dtype, copy=True)
        fn = self._fn("matvec_rows", suf)
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int, ct, ct, C.POINTER(ct), C.c_size_t, C.c_size_t, C.POINTER(ct), C.POINTER(ct), C.POINTER(ct), ct, ct, ct,
                       C.c_size_t, C.c_size_t]
        fn(self._kt(kernel), degree, gamma, coef0, _ptr(X, ct), N, d, _ptr(q, ct), _ptr(dvec, ct), _ptr(out, ct), QA_cost, cost, add,
           row_begin, row_end)
        return out

    def matvec_sampled_rows(self, kernel, X, q, dvec, QA_cost, cost, add, rows, degree=3, gamma=1.0, coef0=0.0):
        """(reference library only) rows `rows` of add * Abar * dvec: the reference's per-pair expression summed over one row each (ref_shim.cpp, sampled_rows)."""
        X = np.ascontiguousarray(X)
        suf, ct = _suffix(X.dtype)
        N, d = X.shape
        q = np.ascontiguousarray(q, dtype=X.dtype)
        dvec = np.ascontiguousarray(dvec, dtype=X.dtype)
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        out = np.zeros(rows.size, dtype=X.dtype)
        fn = self._fn("matvec_sampled_rows", suf)
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int, ct, ct, C.POINTER(ct), C.c_size_t, C.c_size_t, C.POINTER(ct), C.POINTER(ct), ct, ct, ct, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(ct)]
        fn(self._kt(kernel), degree, gamma, coef0, _ptr(X, ct), N, d, _ptr(q, ct), _ptr(dvec, ct), QA_cost, cost, add, rows.ctypes.data_as(C.POINTER(C.c_uint64)), rows.size, _ptr(out, ct))
        return out

    def solve(self, kernel, X, y, eps, max_iter, degree=3, gamma=1.0, coef0=0.0, cost=1.0, trace=False):
        """Returns (alpha[N], rho, info dict[, delta_trace])."""
        X = np.ascontiguousarray(X)
        suf, ct = _suffix(X.dtype)
        N, d = X.shape
        y = np.ascontiguousarray(y, dtype=X.dtype)
        alpha = np.zeros(N, dtype=X.dtype)
        rho = ct(0)
        info = CgInfo()
        cap = int(max_iter) if trace else 0
        tr = np.zeros(max(cap, 1), dtype=np.float64)
        fn = self._fn("solve", suf)
        fn.restype = C.c_int
        fn.argtypes = [C.c_int, C.c_int, ct, ct, ct, C.POINTER(ct), C.c_size_t, C.c_size_t, C.POINTER(ct), ct, C.c_uint64, C.POINTER(ct),
                       C.POINTER(ct), C.POINTER(CgInfo), C.POINTER(C.c_double), C.c_size_t]
        rc = fn(self._kt(kernel), degree, gamma, coef0, cost, _ptr(X, ct), N, d, _ptr(y, ct), eps, int(max_iter), _ptr(alpha, ct),
                C.byref(rho), C.byref(info), _ptr(tr, C.c_double) if trace else None, cap)
        if rc != 0:
            raise RuntimeError(f"{self.prefix}_solve_{suf} failed with status {rc}")
        d_info = {k: getattr(info, k) for k, _ in CgInfo._fields_}
        if trace:
            return alpha, X.dtype.type(rho.value), d_info, tr[:int(d_info["iterations"])].copy()
        return alpha, X.dtype.type(rho.value), d_info

    def calculate_w(self, sv, alpha):
        sv = np.ascontiguousarray(sv)
        suf, ct = _suffix(sv.dtype)
        alpha = np.ascontiguousarray(alpha, dtype=sv.dtype)
        w = np.zeros(sv.shape[1], dtype=sv.dtype)
        fn = self._fn("calculate_w", suf)
        fn.restype = None
        fn.argtypes = [C.POINTER(ct), C.c_size_t, C.c_size_t, C.POINTER(ct), C.POINTER(ct)]
        fn(_ptr(sv, ct), sv.shape[0], sv.shape[1], _ptr(alpha, ct), _ptr(w, ct))
        return w

    def predict_values(self, kernel, sv, alpha, rho, points, w=None, degree=3, gamma=1.0, coef0=0.0):
        """Returns (out[npoints], w or None)."""
        sv = np.ascontiguousarray(sv)
        suf, ct = _suffix(sv.dtype)
        alpha = np.ascontiguousarray(alpha, dtype=sv.dtype)
        points = np.ascontiguousarray(points, dtype=sv.dtype)
        d = sv.shape[1]
        w_valid = C.c_int(0 if w is None else 1)
        w_buf = np.zeros(d, dtype=sv.dtype) if w is None else np.array(w, dtype=sv.dtype, copy=True)
        out = np.zeros(points.shape[0], dtype=sv.dtype)
        fn = self._fn("predict_values", suf)
        fn.restype = None
        fn.argtypes = [C.c_int, C.c_int, ct, ct, C.POINTER(ct), C.c_size_t, C.c_size_t, C.POINTER(ct), ct, C.POINTER(ct), C.POINTER(C.c_int),
                       C.POINTER(ct), C.c_size_t, C.POINTER(ct)]
        fn(self._kt(kernel), degree, gamma, coef0, _ptr(sv, ct), sv.shape[0], d, _ptr(alpha, ct), rho, _ptr(w_buf, ct), C.byref(w_valid),
           _ptr(points, ct), points.shape[0], _ptr(out, ct))
        return out, (w_buf if w_valid.value else None)

    def num_threads(self):
        if self.prefix != "oracle":
            return os.cpu_count()
        self.lib.oracle_num_threads.restype = C.c_int
        return int(self.lib.oracle_num_threads())


_cache = {}


def oracle() -> CpuPath:
    if "oracle" not in _cache:
        if not os.path.isfile(ORACLE_SO):
            raise FileNotFoundError(f"{ORACLE_SO} missing: run `make -C oracle` (or __graft_entry__.build())")
        _cache["oracle"] = CpuPath(ORACLE_SO, "oracle")
    return _cache["oracle"]


def have_ref() -> bool:
    return os.path.isfile(REF_SO)


def ref() -> CpuPath:
    if "ref" not in _cache:
        _cache["ref"] = CpuPath(REF_SO, "ref")
    return _cache["ref"]


def have_ref_release() -> bool:
    return os.path.isfile(REF_RELEASE_SO)


def ref_release() -> CpuPath:
    if "ref_release" not in _cache:
        _cache["ref_release"] = CpuPath(REF_RELEASE_SO, "ref")
    return _cache["ref_release"]


def float_near(a, b, factor=128.0):
    """EXPECT_FLOATING_POINT_VECTOR_NEAR of the reference (tests/custom_test_macros.hpp:114-137, 147-153):
    |a-b| < max(min_normal, factor * eps * (|a|+|b|)) element-wise (or exactly equal)."""
    a = np.asarray(a)
    b = np.asarray(b, dtype=a.dtype)
    fi = np.finfo(a.dtype)
    tol = np.maximum(fi.tiny, factor * fi.eps * (np.abs(a) + np.abs(b)))
    return bool(np.all((a == b) | (np.abs(a - b) < tol)))


def rel_inf(a, b):
    """relative infinity-norm distance ||a-b||_inf / ||b||_inf (the north_star's alpha criterion)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), np.finfo(np.float64).tiny))
