"""Thin Python mirror of the backend boundary over the C ABI (include/plssvm_amd.h).

Function names and argument meaning follow the reference's backend interface so that the parity tests read like the
reference's own (tests/backends/generic_csvm_tests.hpp):

    solve_system_of_linear_equations(params, A, b, eps, max_iter)   csvm.hpp:188-192
    predict_values(params, support_vectors, alpha, rho, w, points)  csvm.hpp:204-208
    generate_q(params, data)                                        gpu_csvm.hpp:349-384 / OpenMP csvm.cpp:232-251
    run_device_kernel(params, q, ret, d, data, QA_cost, add)        gpu_csvm.hpp:431-447 / OpenMP csvm.cpp:283-306
    calculate_w(support_vectors, alpha)                             gpu_csvm.hpp:386-429 / OpenMP csvm.cpp:255-280

All arithmetic runs in the HIP library; numpy is only the host container.
"""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import LssvmCgInfo, LssvmParams, LssvmPredictInfo, LssvmShard, Options, check, ctype_of, lib, options_ptr, ptr, suffix_of
from .exceptions import InvalidParameterError
from .parameter import Parameter

__all__ = ["Options", "Predictor", "solve_system_of_linear_equations", "predict_values", "generate_q", "run_device_kernel", "calculate_w", "ResidentProblem",
           "comm_get_unique_id", "comm_init", "comm_destroy"]


def _params_struct(params: Parameter, num_features: int) -> LssvmParams:
    p = params.resolved(num_features)
    return LssvmParams(int(p.kernel_type), int(p.degree), float(p.gamma), float(p.coef0), float(p.cost))


def _as_matrix(A, dtype=None) -> np.ndarray:
    A = np.asarray(A)
    if dtype is None:
        dtype = A.dtype if A.dtype in (np.float32, np.float64) else np.float64
    A = np.ascontiguousarray(A, dtype=dtype)
    if A.ndim != 2:
        raise InvalidParameterError("All data points must have the same number of features!")
    return A


def solve_system_of_linear_equations(params: Parameter, A, b, eps: float, max_iter: int, devices=None, num_devices: int | None = None, options: Options | None = None):
    """Returns ``(alpha[N], rho, info)`` -- ``csvm::solve_system_of_linear_equations`` (csvm.hpp:188-192).

    ``devices`` (a list of HIP ordinals; the same ordinal may repeat) or ``num_devices`` (0 = every visible device) select the
    single-process multi-device solve ``lssvm_mi355_solve_multi_*``; with neither the solve runs on device 0.  ``options``: this call's own tuning knobs
    (:class:`Options`); None = the process defaults."""
    A = _as_matrix(A)
    N, d = A.shape
    b = np.ascontiguousarray(b, dtype=A.dtype)
    if b.shape != (N,):
        raise InvalidParameterError(f"The number of data points in the matrix A ({N}) and the values in the right hand side vector ({b.size}) must be the same!")
    ct = ctype_of(A.dtype)
    alpha = np.zeros(N, dtype=A.dtype)
    rho = ct(0)
    info = LssvmCgInfo()
    ps = _params_struct(params, d)
    if devices is None and num_devices is None:
        fn = getattr(lib, f"lssvm_mi355_solve_{suffix_of(A.dtype)}")
        fn.restype = C.c_int
        check(fn(C.byref(ps), ptr(A), C.c_size_t(N), C.c_size_t(d), ptr(b), ct(eps), C.c_uint64(int(max_iter)), ptr(alpha), C.byref(rho), C.byref(info), options_ptr(options)))
    else:
        fn = getattr(lib, f"lssvm_mi355_solve_multi_{suffix_of(A.dtype)}")
        fn.restype = C.c_int
        dev_arr, ndev = _capi.int_array(devices)
        if devices is None:
            ndev = int(num_devices)
        check(fn(C.byref(ps), ptr(A), C.c_size_t(N), C.c_size_t(d), ptr(b), ct(eps), C.c_uint64(int(max_iter)), ptr(alpha), C.byref(rho), C.byref(info),
                 dev_arr, C.c_int(ndev), options_ptr(options)))
    return alpha, A.dtype.type(rho.value), info.as_dict()


def generate_q(params: Parameter, data, options: Options | None = None):
    data = _as_matrix(data)
    N, d = data.shape
    q = np.zeros(N - 1, dtype=data.dtype)
    fn = getattr(lib, f"lssvm_mi355_generate_q_{suffix_of(data.dtype)}")
    fn.restype = C.c_int
    ps = _params_struct(params, d)
    check(fn(C.byref(ps), ptr(data), C.c_size_t(N), C.c_size_t(d), ptr(q), options_ptr(options)))
    return q


def run_device_kernel(params: Parameter, q, ret, d, data, QA_cost: float, add: float, options: Options | None = None):
    """``ret += add * Abar * d``; returns the updated copy of ``ret`` (gpu_csvm.hpp:431-447)."""
    data = _as_matrix(data)
    N, nf = data.shape
    q = np.ascontiguousarray(q, dtype=data.dtype)
    d = np.ascontiguousarray(d, dtype=data.dtype)
    out = np.array(ret, dtype=data.dtype, copy=True)
    if not (q.size == d.size == out.size == N - 1):
        raise InvalidParameterError(f"Sizes mismatch!: {q.size} != {N - 1}")
    ct = ctype_of(data.dtype)
    fn = getattr(lib, f"lssvm_mi355_run_device_kernel_{suffix_of(data.dtype)}")
    fn.restype = C.c_int
    ps = _params_struct(params, nf)
    check(fn(C.byref(ps), ptr(data), C.c_size_t(N), C.c_size_t(nf), ptr(q), ptr(d), ptr(out), ct(QA_cost), ct(add), options_ptr(options)))
    return out


def calculate_w(support_vectors, alpha):
    sv = _as_matrix(support_vectors)
    alpha = np.ascontiguousarray(alpha, dtype=sv.dtype)
    if alpha.size != sv.shape[0]:
        raise InvalidParameterError(f"The number of support vectors ({sv.shape[0]}) and weights ({alpha.size}) must match!")
    w = np.zeros(sv.shape[1], dtype=sv.dtype)
    fn = getattr(lib, f"lssvm_mi355_calculate_w_{suffix_of(sv.dtype)}")
    fn.restype = C.c_int
    check(fn(ptr(sv), C.c_size_t(sv.shape[0]), C.c_size_t(sv.shape[1]), ptr(alpha), ptr(w)))
    return w


def predict_values(params: Parameter, support_vectors, alpha, rho: float, w, predict_points, options: Options | None = None, info_out: dict | None = None):
    """Returns ``(values[num_points], w)``; ``w`` is None for the polynomial / rbf kernels (csvm.hpp:204-208).  ``info_out``: a dict that receives the call's
    ``lssvm_predict_info`` (timings, the Gram mode that ran)."""
    sv = _as_matrix(support_vectors)
    pts = _as_matrix(predict_points, dtype=sv.dtype)
    alpha = np.ascontiguousarray(alpha, dtype=sv.dtype)
    nsv, nf = sv.shape
    if alpha.size != nsv:
        raise InvalidParameterError(f"The number of support vectors ({nsv}) and number of weights ({alpha.size}) must be the same!")
    if pts.shape[1] != nf:
        raise InvalidParameterError(f"The number of features in the support vectors ({nf}) must be the same as in the data points to predict ({pts.shape[1]})!")
    if w is not None and len(w) not in (0, nf):
        raise InvalidParameterError(f"Either w must be empty or contain exactly the same number of values ({len(w)}) as features are present ({nf})!")
    ct = ctype_of(sv.dtype)
    w_valid = C.c_int(1 if (w is not None and len(w) == nf) else 0)
    w_buf = np.array(w, dtype=sv.dtype, copy=True) if w_valid.value else np.zeros(nf, dtype=sv.dtype)
    out = np.zeros(pts.shape[0], dtype=sv.dtype)
    fn = getattr(lib, f"lssvm_mi355_predict_values_{suffix_of(sv.dtype)}")
    fn.restype = C.c_int
    ps = _params_struct(params, nf)
    pinfo = LssvmPredictInfo()
    check(fn(C.byref(ps), ptr(sv), C.c_size_t(nsv), C.c_size_t(nf), ptr(alpha), ct(rho), ptr(w_buf), C.byref(w_valid), ptr(pts), C.c_size_t(pts.shape[0]), ptr(out), C.byref(pinfo), options_ptr(options)))
    if info_out is not None:
        info_out.update(pinfo.as_dict())
    return out, (w_buf if w_valid.value else None)


class Predictor:
    """A model resident in HBM across predict calls (``lssvm_mi355_predictor_*``): the support vectors are uploaded and prepared once, every :meth:`predict` uploads only
    its batch of points.  Same values as :func:`predict_values`; ``info_out["resident"]`` says whether a batch ran against the resident form or took the one-shot path."""

    def __init__(self, params: Parameter, support_vectors, alpha, rho: float, options: Options | None = None):
        sv = _as_matrix(support_vectors)
        alpha = np.ascontiguousarray(alpha, dtype=sv.dtype)
        if alpha.size != sv.shape[0]:
            raise InvalidParameterError(f"The number of support vectors ({sv.shape[0]}) and number of weights ({alpha.size}) must be the same!")
        self.dtype, self.num_features = sv.dtype, int(sv.shape[1])
        self._h = C.c_void_p(None)
        ps = _params_struct(params, sv.shape[1])
        check(lib.lssvm_mi355_predictor_create(C.byref(self._h), C.byref(ps), C.c_int(_capi.dtype_code(sv.dtype)), ptr(sv), C.c_size_t(sv.shape[0]), C.c_size_t(sv.shape[1]), ptr(alpha),
                                               C.c_double(float(rho)), options_ptr(options)))

    def predict(self, predict_points, info_out: dict | None = None):
        pts = _as_matrix(predict_points, dtype=self.dtype)
        if pts.shape[1] != self.num_features:
            raise InvalidParameterError(f"The number of features in the support vectors ({self.num_features}) must be the same as in the data points to predict ({pts.shape[1]})!")
        out = np.zeros(pts.shape[0], dtype=self.dtype)
        pinfo = LssvmPredictInfo()
        check(lib.lssvm_mi355_predictor_predict(self._h, ptr(pts), C.c_int(_capi.LSSVM_MEM_HOST), C.c_size_t(pts.shape[0]), ptr(out), C.byref(pinfo)))
        if info_out is not None:
            info_out.update(pinfo.as_dict())
        return out

    def predict_device(self, points_ptr: int, num_points: int, out_ptr: int, info_out: dict | None = None) -> None:
        """The same with the batch AND the values in memory of device 0 (``LSSVM_MEM_DEVICE``): ``points_ptr`` -> ``num_points`` x ``num_features`` row-major of the predictor's
        dtype (e.g. a contiguous torch tensor's ``data_ptr()``, complete when the call is made), ``out_ptr`` -> ``num_points`` values, written when the call returns."""
        pinfo = LssvmPredictInfo()
        check(lib.lssvm_mi355_predictor_predict(self._h, C.c_void_p(int(points_ptr)), C.c_int(_capi.LSSVM_MEM_DEVICE), C.c_size_t(int(num_points)), C.c_void_p(int(out_ptr)), C.byref(pinfo)))
        if info_out is not None:
            info_out.update(pinfo.as_dict())

    def close(self):
        if self._h:
            lib.lssvm_mi355_predictor_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


# ---------------------------------------------------------------------------------------------------------------------
# communicator + resident problem (bench.py, multi-rank launchers)
# ---------------------------------------------------------------------------------------------------------------------
def comm_get_unique_id() -> bytes:
    buf = (C.c_ubyte * _capi.UNIQUE_ID_BYTES)()
    check(lib.lssvm_mi355_comm_get_unique_id(buf))
    return bytes(buf)


def comm_init(device: int, rank: int, world: int, unique_id: bytes) -> None:
    if len(unique_id) != _capi.UNIQUE_ID_BYTES:
        raise InvalidParameterError("unique id must have 128 bytes")
    buf = (C.c_ubyte * _capi.UNIQUE_ID_BYTES).from_buffer_copy(unique_id)
    check(lib.lssvm_mi355_comm_init(C.c_int(device), C.c_int(rank), C.c_int(world), buf))


def comm_library_path() -> str:
    """The file the library resolved its RCCL entry points from (a measurement / test aid, include/plssvm_amd_testing.h)."""
    buf = C.create_string_buffer(4096)
    check(lib.lssvm_mi355_comm_library_path(buf, C.c_size_t(4096)))
    return buf.value.decode()


def comm_destroy() -> None:
    check(lib.lssvm_mi355_comm_destroy())


class ResidentProblem:
    """A data matrix resident in HBM plus the CG state on it (``lssvm_mi355_problem_*`` / ``lssvm_mi355_cg_*``)."""

    def __init__(self, params: Parameter, X, device: int = 0, rank: int = 0, world: int = 1, device_ptr: int | None = None, shape=None, dtype=None,
                 devices=None, options: Options | None = None):
        """``devices``: a list of HIP ordinals -> ONE process drives all of them (``lssvm_mi355_problem_create_multi``; an empty list = every
        visible device); otherwise ``device`` holds rank ``rank`` of a world of processes (one process per GPU, or a single GPU)."""
        self._h = C.c_void_p(None)
        if device_ptr is not None:
            N, d = shape
            self.dtype = np.dtype(dtype)
            src = C.c_void_p(device_ptr)
            kind = _capi.LSSVM_MEM_DEVICE
            self._keep = None
        else:
            X = _as_matrix(X)
            N, d = X.shape
            self.dtype = X.dtype
            src = ptr(X)
            kind = _capi.LSSVM_MEM_HOST
            self._keep = X
        self.num_points, self.num_features = int(N), int(d)
        self.params = params.resolved(d)
        ps = _params_struct(params, d)
        if devices is not None:
            if (rank, world) != (0, 1):
                raise InvalidParameterError("a device list (one process, several GPUs) and rank/world (one process per GPU) exclude each other")
            dev_arr, ndev = _capi.int_array(list(devices) or None)
            lib.lssvm_mi355_problem_create_multi.restype = C.c_int
            check(lib.lssvm_mi355_problem_create_multi(C.byref(self._h), C.byref(ps), C.c_int(_capi.dtype_code(self.dtype)), src, C.c_int(kind), C.c_size_t(N),
                                                       C.c_size_t(d), dev_arr, C.c_int(ndev), options_ptr(options)))
        else:
            shard = LssvmShard(rank, world)
            lib.lssvm_mi355_problem_create.restype = C.c_int
            check(lib.lssvm_mi355_problem_create(C.byref(self._h), C.byref(ps), C.c_int(_capi.dtype_code(self.dtype)), src, C.c_int(kind), C.c_size_t(N), C.c_size_t(d),
                                                 C.c_int(device), C.byref(shard), options_ptr(options)))
        self._keep = None  # the library copied the data

    def close(self):
        if self._h:
            lib.lssvm_mi355_problem_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def q(self):
        q = np.zeros(self.num_points - 1, dtype=self.dtype)
        qa = C.c_double(0)
        check(lib.lssvm_mi355_problem_get_q(self._h, ptr(q), C.byref(qa)))
        return q, float(qa.value)

    def matvec(self, d, ret, add: float = 1.0):
        d = np.ascontiguousarray(d, dtype=self.dtype)
        out = np.array(ret, dtype=self.dtype, copy=True)
        if not (d.ndim == out.ndim == 1 and d.size == out.size == self.num_points - 1):  # the library reads / writes exactly N - 1 elements of both
            raise InvalidParameterError(f"Sizes mismatch!: {d.size} / {out.size} != {self.num_points - 1}")
        check(lib.lssvm_mi355_problem_matvec(self._h, ptr(d), ptr(out), C.c_double(add)))
        return out

    def cg_begin(self, y, eps: float):
        y = np.ascontiguousarray(y, dtype=self.dtype)
        if y.size != self.num_points:
            raise InvalidParameterError(f"The number of data points in the matrix A ({self.num_points}) and the values in the right hand side vector ({y.size}) must be the same!")
        check(lib.lssvm_mi355_cg_begin(self._h, ptr(y), C.c_double(eps)))

    def cg_step(self, iterations: int) -> bool:
        done = C.c_int(0)
        check(lib.lssvm_mi355_cg_step(self._h, C.c_uint64(int(iterations)), C.byref(done)))
        return bool(done.value)

    def cg_finish(self):
        alpha = np.zeros(self.num_points, dtype=self.dtype)
        rho = C.c_double(0)
        info = LssvmCgInfo()
        check(lib.lssvm_mi355_cg_finish(self._h, ptr(alpha), C.byref(rho), C.byref(info)))
        return alpha, self.dtype.type(rho.value), info.as_dict()

    def info(self):
        info = LssvmCgInfo()
        check(lib.lssvm_mi355_problem_info(self._h, C.byref(info)))
        return info.as_dict()

    def synchronize(self):
        check(lib.lssvm_mi355_problem_synchronize(self._h))

    def rebalance(self, weights=None) -> bool:
        """New shares for the ranks of a sharded symmetric problem, between two ``cg_step`` calls (``lssvm_mi355_problem_rebalance``): explicit ``weights`` (one per rank,
        the same on every rank) or, ``None``, by the shards' measured pace.  Returns whether the shares changed."""
        w = [float(v) for v in (weights or [])]
        arr = (C.c_double * len(w))(*w) if w else None
        changed = C.c_int(0)
        check(lib.lssvm_mi355_problem_rebalance(self._h, arr, C.c_int(len(w)), C.byref(changed)))
        return bool(changed.value)

    def ipc_export(self) -> bytes:
        """One process per GPU over HIP IPC: this rank's ``LSSVM_IPC_BLOB_BYTES`` blob (``lssvm_mi355_problem_ipc_export``)."""
        buf = (C.c_ubyte * _capi.LSSVM_IPC_BLOB_BYTES)()
        check(lib.lssvm_mi355_problem_ipc_export(self._h, buf, C.c_size_t(_capi.LSSVM_IPC_BLOB_BYTES)))
        return bytes(buf)

    def ipc_connect(self, blobs) -> None:
        """``blobs``: the exports of ALL ranks in rank order (``lssvm_mi355_problem_ipc_connect``)."""
        joined = b"".join(bytes(b) for b in blobs)
        buf = (C.c_ubyte * len(joined)).from_buffer_copy(joined) if joined else None
        check(lib.lssvm_mi355_problem_ipc_connect(self._h, buf, C.c_size_t(len(joined))))
