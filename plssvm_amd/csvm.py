"""``CSVM`` -- the Python mirror of ``plssvm::csvm`` and of the reference's Python bindings
(bindings/Python/csvm.cpp:26-52, :95-181): ``fit`` / ``predict`` / ``score`` on top of the backend boundary, plus
``make_csvm`` (csvm_factory.hpp:123-171) with this repository's single backend."""

from __future__ import annotations

import enum
import sys
import time

import numpy as np

from . import _capi, backend
from .data_set import DataSet
from .exceptions import BackendError, InvalidParameterError, UnsupportedBackendError
from .model import Model
from .parameter import KernelFunctionType, Parameter

__all__ = ["CSVM", "MI355CSVM", "make_csvm", "BackendType", "TargetPlatform"]


class BackendType(enum.Enum):
    """plssvm::backend_type (backend_types.hpp:30-43) + ``mi355``."""
    AUTOMATIC = "automatic"
    OPENMP = "openmp"
    CUDA = "cuda"
    HIP = "hip"
    OPENCL = "opencl"
    SYCL = "sycl"
    MI355 = "mi355"


class TargetPlatform(enum.Enum):
    AUTOMATIC = "automatic"
    CPU = "cpu"
    GPU_NVIDIA = "gpu_nvidia"
    GPU_AMD = "gpu_amd"
    GPU_INTEL = "gpu_intel"


class CSVM:
    """Abstract base: subclasses implement the two boundary methods (csvm.hpp:188-208)."""

    def __init__(self, params: Parameter | None = None, **kwargs):
        if params is not None and kwargs:
            raise InvalidParameterError("Provide either a Parameter object or keyword arguments, not both!")
        self.params = params if params is not None else Parameter(**kwargs)
        self.target_platform = TargetPlatform.AUTOMATIC
        self.last_cg_info = None

    # --- boundary (pure virtual in the reference) ---
    def solve_system_of_linear_equations(self, params, A, b, eps, max_iter):
        raise NotImplementedError

    def predict_values(self, params, support_vectors, alpha, rho, w, predict_points):
        raise NotImplementedError

    def get_params(self):
        return self.params

    def set_params(self, params: Parameter | None = None, **kwargs):
        """csvm::set_params (csvm.hpp:236-261): named arguments override only what they name."""
        if params is not None:
            self.params = params
        for k, v in kwargs.items():
            if k not in ("kernel_type", "degree", "gamma", "coef0", "cost"):
                raise InvalidParameterError(f"Invalid argument {k} provided!")
            setattr(self.params, k, v)
        self.params.__post_init__()

    def get_target_platform(self):
        return self.target_platform

    # --- fit / predict / score (csvm.hpp:263-375) ---
    def fit(self, data: DataSet, epsilon: float = 0.001, max_iter: int | None = None) -> Model:
        if epsilon <= 0.0:
            raise InvalidParameterError(f"epsilon must be less than 0.0, but is {epsilon}!")  # csvm.hpp:283 (message verbatim)
        if max_iter is not None and max_iter <= 0:
            raise InvalidParameterError(f"max_iter must be greater than 0, but is {max_iter}!")  # csvm.hpp:291
        if not data.has_labels():
            raise InvalidParameterError("No labels given for training! Maybe the data is only usable for prediction?")  # csvm.hpp:298
        if max_iter is None:
            max_iter = data.num_data_points()  # csvm.hpp:269
        params = self.params.resolved(data.num_features())  # csvm.hpp:303-307
        t0 = time.perf_counter()
        alpha, rho, info = self.solve_system_of_linear_equations(params, data.data(), data.mapped_labels(), epsilon, max_iter)
        info["total_runtime_ms"] = (time.perf_counter() - t0) * 1e3  # cg/total_runtime (csvm.hpp:318-320)
        self.last_cg_info = info
        return Model(params, data, alpha=alpha, rho=rho)

    def predict(self, model: Model, data: DataSet):
        if model.num_features() != data.num_features():
            raise InvalidParameterError(f"Number of features per data point ({data.num_features()}) must match the number of features per support vector of the "
                                        f"provided model ({model.num_features()})!")
        values, w = self.predict_values(model.params, model.support_vectors(), model.alpha, float(model.rho), model.w, data.data())
        if w is not None:
            model.w = w
        mapper = model.data.mapping
        return [mapper.label_of(1 if v > 0 else -1) for v in values]  # operators.hpp:180-182 sign, csvm.hpp:337-340

    def score(self, model: Model, data: DataSet | None = None) -> float:
        data = model.data if data is None else data
        if not data.has_labels():
            raise InvalidParameterError("The data set to score must have labels!")
        if model.num_features() != data.num_features():
            raise InvalidParameterError(f"Number of features per data point ({data.num_features()}) must match the number of features per support vector of the "
                                        f"provided model ({model.num_features()})!")
        predicted = self.predict(model, data)
        correct = sum(1 for p, c in zip(predicted, data.labels()) if p == c)
        return correct / len(predicted)


class MI355CSVM(CSVM):
    """The MI355X backend (counterpart of plssvm::hip::csvm, HIP/csvm.hpp:39-99, csvm.hip.cpp:47-85)."""

    def __init__(self, target=TargetPlatform.AUTOMATIC, params: Parameter | None = None, num_devices: int = 1, **kwargs):
        """``num_devices``: devices ONE solve is sharded over -- 1 (default) = device 0 only, k = devices 0 .. k-1 (gpu_csvm.hpp:283-299),
        0 = automatic (every visible device, at least 4096 points each, as the reference's backends take every device they find,
        csvm.hip.cpp:66-75).  Several devices are OPT-IN: results then depend on the device count through the order of the sums, the exchange
        bootstraps RCCL inside the process, and that path has not run on a multi-GPU box yet."""
        if isinstance(target, Parameter):
            target, params = TargetPlatform.AUTOMATIC, target
        super().__init__(params, **kwargs)
        if target not in (TargetPlatform.AUTOMATIC, TargetPlatform.GPU_AMD):
            raise BackendError(f"Invalid target platform '{target.value}' for the MI355 backend!")
        self.target_platform = TargetPlatform.GPU_AMD
        self.num_devices = _capi.device_count()
        if self.num_devices <= 0:
            raise BackendError("MI355 backend selected but no HIP capable devices were found!")
        if not 0 <= int(num_devices) <= self.num_devices:
            raise BackendError(f"Requested {num_devices} devices, but only {self.num_devices} are available!")
        self.use_devices = int(num_devices)
        self._options = None  # this object's own tuning knobs (ABI 4): created by the first set_option, None = the process defaults

    def set_option(self, name: str, value: int) -> None:
        """A tuning knob of THIS backend object (``lssvm_mi355_options``; names as ``lssvm_mi355_set_option``): other objects and the process defaults are not touched --
        the reference's backend objects share no state beyond ``verbosity`` either (csvm.hpp:50-83)."""
        if self._options is None:
            self._options = _capi.Options()
        self._options.set(name, value)

    def get_option(self, name: str) -> int:
        return self._options.get(name) if self._options is not None else _capi.get_option(name)

    def solve_system_of_linear_equations(self, params, A, b, eps, max_iter):
        # all devices of this process behind ONE call (gpu_csvm::solve_system_of_linear_equations_impl, gpu_csvm.hpp:477-654)
        return backend.solve_system_of_linear_equations(params, A, b, eps, max_iter, num_devices=self.use_devices, options=self._options)

    def predict_values(self, params, support_vectors, alpha, rho, w, predict_points):
        return backend.predict_values(params, support_vectors, alpha, rho, w, predict_points, options=self._options)

    def predict(self, model: Model, data: DataSet):
        """csvm::predict (csvm.hpp:322-342) with the model RESIDENT in HBM from the first call on (``lssvm_mi355_predictor_*``): later calls with the same model upload only
        their points.  Same labels as the base class's one-shot ``predict_values``."""
        if model.num_features() != data.num_features():
            raise InvalidParameterError(f"Number of features per data point ({data.num_features()}) must match the number of features per support vector of the "
                                        f"provided model ({model.num_features()})!")
        t0 = time.perf_counter()
        cached = getattr(model, "_predictor", None)
        if cached is None or cached[0] is not self:
            cached = (self, backend.Predictor(model.params, model.support_vectors(), model.alpha, float(model.rho), options=self._options))
            model._predictor = cached
        t1 = time.perf_counter()
        info = {}
        values = cached[1].predict(data.data(), info_out=info)
        t2 = time.perf_counter()
        mapper = model.data.mapping
        pos, neg = mapper.label_of(1), mapper.label_of(-1)
        labels = [pos if p else neg for p in (np.asarray(values) > 0).tolist()]  # operators.hpp:180-182 sign, csvm.hpp:337-340
        # where the call's time went, in seconds (the command line's timing block and bench.py's `e2e` print it)
        self.last_predict_phases = {"model_to_hbm_s": t1 - t0, "values_s": t2 - t1, "library_total_ms": float(info.get("total_ms", 0.0)), "kernel_ms": float(info.get("kernel_ms", 0.0)),
                                    "labels_s": time.perf_counter() - t2}
        return labels


def make_csvm(backend_type=BackendType.AUTOMATIC, *args, **kwargs) -> CSVM:
    """plssvm::make_csvm (csvm_factory.hpp:123-171).  ``automatic`` / ``mi355`` / ``hip`` select the MI355X backend; every other
    enumerator raises UnsupportedBackendError("No {} backend available!") like a reference build without that backend."""
    if isinstance(backend_type, str):
        try:
            backend_type = BackendType(backend_type.lower())
        except ValueError:
            raise UnsupportedBackendError("Unrecognized backend provided!") from None
    if not isinstance(backend_type, BackendType):
        args = (backend_type,) + args
        backend_type = BackendType.AUTOMATIC
    if backend_type in (BackendType.AUTOMATIC, BackendType.MI355, BackendType.HIP):
        return MI355CSVM(*args, **kwargs)
    raise UnsupportedBackendError(f"No {backend_type.value} backend available!")
