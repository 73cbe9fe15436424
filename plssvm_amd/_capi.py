"""ctypes binding of the C ABI declared in ``include/plssvm_amd.h`` (libplssvm_amd.so: HIP kernels for gfx950).

The library is the ONLY compute path of this package: if it is missing, importing this module raises -- there is no
CPU or PyTorch fallback (the CPU oracle under ``oracle/`` is test infrastructure and is never imported from here).
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .exceptions import BackendError, InvalidParameterError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PLSSVM_AMD_LIBRARY") or os.path.join(_HERE, "lib", "libplssvm_amd.so")  # (the variable lets a developer A/B two builds)

LSSVM_DTYPE_F32 = 0
LSSVM_DTYPE_F64 = 1
LSSVM_MEM_HOST = 0
LSSVM_MEM_DEVICE = 1
UNIQUE_ID_BYTES = 128

STATUS_NAMES = {0: "LSSVM_SUCCESS", -1: "LSSVM_ERR_INVALID_ARGUMENT", -2: "LSSVM_ERR_NO_DEVICE", -3: "LSSVM_ERR_HIP", -4: "LSSVM_ERR_COMM",
                -5: "LSSVM_ERR_OUT_OF_MEMORY", -6: "LSSVM_ERR_INTERNAL"}

# every symbol include/plssvm_amd.h declares (tests/test_capi_symbols.py checks the built library against this list AND the header)
EXPORTED_SYMBOLS = [
    "lssvm_mi355_abi_version", "lssvm_mi355_device_count", "lssvm_mi355_device_name", "lssvm_mi355_last_error",
    "lssvm_mi355_options_create", "lssvm_mi355_options_set", "lssvm_mi355_options_get", "lssvm_mi355_options_destroy",
    "lssvm_mi355_solve_f32", "lssvm_mi355_solve_f64", "lssvm_mi355_solve_multi_f32", "lssvm_mi355_solve_multi_f64", "lssvm_mi355_predict_values_f32", "lssvm_mi355_predict_values_f64",
    "lssvm_mi355_predictor_create", "lssvm_mi355_predictor_predict", "lssvm_mi355_predictor_destroy",
    "lssvm_mi355_generate_q_f32", "lssvm_mi355_generate_q_f64", "lssvm_mi355_run_device_kernel_f32", "lssvm_mi355_run_device_kernel_f64",
    "lssvm_mi355_calculate_w_f32", "lssvm_mi355_calculate_w_f64",
    "lssvm_mi355_shard_blocks", "lssvm_mi355_set_shard_weights", "lssvm_mi355_problem_rebalance", "lssvm_mi355_comm_get_unique_id", "lssvm_mi355_comm_init", "lssvm_mi355_comm_destroy",
    "lssvm_mi355_problem_create", "lssvm_mi355_problem_create_multi", "lssvm_mi355_problem_ipc_export", "lssvm_mi355_problem_ipc_connect", "lssvm_mi355_problem_destroy", "lssvm_mi355_problem_get_q", "lssvm_mi355_problem_matvec",
    "lssvm_mi355_cg_begin", "lssvm_mi355_cg_step", "lssvm_mi355_cg_finish", "lssvm_mi355_problem_synchronize", "lssvm_mi355_problem_info",
    "lssvm_mi355_measure_bf16_mfma_ceiling", "lssvm_mi355_comm_library_path", "lssvm_mi355_set_io_threads", "lssvm_mi355_set_option", "lssvm_mi355_get_option",
    "lssvm_mi355_libsvm_open", "lssvm_mi355_libsvm_fill_f32", "lssvm_mi355_libsvm_fill_f64", "lssvm_mi355_libsvm_close",
    "lssvm_mi355_libsvm_write_f32", "lssvm_mi355_libsvm_write_f64", "lssvm_mi355_model_write_f32", "lssvm_mi355_model_write_f64",
    "lssvm_mi355_model_open", "lssvm_mi355_model_labels", "lssvm_mi355_model_fill_f32", "lssvm_mi355_model_fill_f64", "lssvm_mi355_model_close",
    "lssvm_mi355_arff_open", "lssvm_mi355_arff_fill_f32", "lssvm_mi355_arff_fill_f64", "lssvm_mi355_arff_close",
]


class LssvmParams(C.Structure):
    """``lssvm_params`` (include/plssvm_amd.h) == plssvm::detail::parameter<T> with gamma resolved."""
    _fields_ = [("kernel_type", C.c_int32), ("degree", C.c_int32), ("gamma", C.c_double), ("coef0", C.c_double), ("cost", C.c_double)]


class LssvmCgInfo(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("max_iterations", C.c_uint64), ("residuum", C.c_double), ("initial_residuum", C.c_double),
                ("target_residuum", C.c_double), ("epsilon", C.c_double), ("avg_iteration_ms", C.c_double), ("total_ms", C.c_double),
                ("setup_ms", C.c_double), ("matvec_kernel_ms", C.c_double), ("matvec_launches", C.c_uint64), ("devices_used", C.c_int32),
                ("converged", C.c_int32), ("symmetric", C.c_int32), ("gram_mode", C.c_int32), ("local_devices", C.c_int32), ("exchange", C.c_int32), ("tile_launches_per_matvec", C.c_int32), ("rbf_direct", C.c_int32), ("rbf_exponent_scale", C.c_double),
                ("matvec_timed", C.c_uint64), ("matvec_kernel_ms_total", C.c_double), ("rccl_nranks", C.c_int32), ("rccl_rank", C.c_int32), ("rccl_device", C.c_int32), ("persistent_launches", C.c_int32),
                ("f16_row_rel_error", C.c_double)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class LssvmPredictInfo(C.Structure):
    """``lssvm_predict_info``: the timings of one ``predict_values`` call."""
    _fields_ = [("total_ms", C.c_double), ("setup_ms", C.c_double), ("kernel_ms", C.c_double), ("rbf_exponent_scale", C.c_double), ("f16_row_rel_error", C.c_double),
                ("gram_mode", C.c_int32), ("rbf_direct", C.c_int32), ("resident", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


class LssvmModelInfo(C.Structure):
    """``lssvm_model_info``: the header of a model file as the native reader reports it."""
    _fields_ = [("kernel_type", C.c_int32), ("has_degree", C.c_int32), ("has_gamma", C.c_int32), ("has_coef0", C.c_int32), ("degree", C.c_int64),
                ("gamma", C.c_double), ("coef0", C.c_double), ("rho", C.c_double), ("nr_class", C.c_uint64), ("total_sv", C.c_uint64),
                ("num_features", C.c_uint64), ("label_text_bytes", C.c_uint64)]


class LssvmShard(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32)]


def _load():
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C plssvm_amd/csrc`. plssvm_amd has no CPU fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()
lib.lssvm_mi355_last_error.restype = C.c_char_p
lib.lssvm_mi355_abi_version.restype = C.c_int
lib.lssvm_mi355_device_count.restype = C.c_int


def last_error() -> str:
    msg = lib.lssvm_mi355_last_error()
    return msg.decode("utf-8", errors="replace") if msg else ""


def check(status: int) -> None:
    """Map a non-zero status to the exception the reference would throw at this point."""
    if status == 0:
        return
    msg = f"{STATUS_NAMES.get(status, status)}: {last_error()}"
    if status == -1:
        raise InvalidParameterError(msg)
    raise BackendError(msg)


def dtype_code(dtype) -> int:
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return LSSVM_DTYPE_F32
    if dtype == np.float64:
        return LSSVM_DTYPE_F64
    raise InvalidParameterError(f"real_type must be float32 or float64, not {dtype}")


def ctype_of(dtype):
    return C.c_float if np.dtype(dtype) == np.float32 else C.c_double


def suffix_of(dtype) -> str:
    return "f32" if np.dtype(dtype) == np.float32 else "f64"


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def device_count() -> int:
    return int(lib.lssvm_mi355_device_count())


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    check(lib.lssvm_mi355_device_name(C.c_int(device), buf, C.c_size_t(256)))
    return buf.value.decode()


ABI_VERSION = 4
LSSVM_IPC_BLOB_BYTES = 256
# every tuning knob of lssvm_mi355_set_option (include/plssvm_amd.h)
OPTION_NAMES = ["rbf_form", "rbf_fold", "j_chunk_tiles", "j_chunk_head", "symmetric", "tile_kernel", "gram_mode", "mfma_shape", "colslab_band_mb", "colslab_limit_mb", "force_collective", "skip_collective",
                "exchange", "ipc_timeout_s", "enqueue_ahead_below_us", "rebalance_after"]
# accepted with the value 0 everywhere, with other values in development builds only (make DEV=1 / -DLSSVM_ENABLE_ABLATION)
DEV_OPTION_NAMES = ["pair_lag", "debug_ablate", "item_order_dev"]


def int_array(values):
    """A C ``int[]`` (or NULL for None) for the device lists of the ``_multi`` entry points."""
    if values is None:
        return None, 0
    values = [int(v) for v in values]
    return (C.c_int * len(values))(*values), len(values)


def shard_blocks(num_points: int, world: int, rank: int, symmetric: bool):
    """The library's own row-block partition (host only): ``(block_begin, block_end)`` of shard ``rank``."""
    b, e = C.c_int64(0), C.c_int64(0)
    check(lib.lssvm_mi355_shard_blocks(C.c_size_t(num_points), C.c_int(world), C.c_int(rank), C.c_int(1 if symmetric else 0), C.byref(b), C.byref(e)))
    return int(b.value), int(e.value)


def set_shard_weights(weights=None) -> None:
    """Shares of the triangle's area per rank of a sharded symmetric problem (``None`` / empty: equal shares) -- a process-wide default, snapshotted when a
    problem is created; every rank must set the same list (``lssvm_mi355_set_shard_weights``)."""
    w = [float(v) for v in (weights or [])]
    arr = (C.c_double * len(w))(*w) if w else None
    check(lib.lssvm_mi355_set_shard_weights(arr, C.c_int(len(w))))


class Options:
    """``lssvm_mi355_options`` (ABI 4): the tuning knobs held by ONE caller instead of the process -- a private copy of the process defaults of the moment it is
    created, changed with :meth:`set`, handed to the entry points that create a problem (``options=`` of ``plssvm_amd.backend``)."""

    def __init__(self, **values):
        self._h = C.c_void_p(None)
        check(lib.lssvm_mi355_options_create(C.byref(self._h)))
        for name, value in values.items():
            self.set(name, value)

    def set(self, name: str, value: int) -> "Options":
        check(lib.lssvm_mi355_options_set(self._h, name.encode(), C.c_int64(int(value))))
        return self

    def get(self, name: str) -> int:
        v = C.c_int64(0)
        check(lib.lssvm_mi355_options_get(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def close(self):
        if self._h:
            lib.lssvm_mi355_options_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def options_ptr(options):
    """The ``const lssvm_mi355_options *`` argument: NULL (= the process defaults) for None."""
    if options is None:
        return None
    if not isinstance(options, Options):
        raise InvalidParameterError("options must be a plssvm_amd._capi.Options or None")
    return options._h


def set_option(name: str, value: int) -> None:
    check(lib.lssvm_mi355_set_option(name.encode(), C.c_int64(value)))


def get_option(name: str) -> int:
    v = C.c_int64(0)
    check(lib.lssvm_mi355_get_option(name.encode(), C.byref(v)))
    return int(v.value)
