#!/usr/bin/env python3
"""fp64 on wide data: one CG iteration of the linear kernel (one pass up to 256 features, beyond that one pass per feature panel of 128, or of 64
with the option linear_panel_features = 64), and of rbf / polynomial (generic full-square kernel beyond 256 features)."""
import time

import numpy as np

from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter


def iteration_ms(p, X, y):
    with backend.ResidentProblem(p, X) as prob:
        prob.cg_begin(y, 1e-30)
        prob.cg_step(3)
        t0 = time.perf_counter()
        prob.cg_step(5)
        info = prob.cg_finish()[2]
        return 1e3 * (time.perf_counter() - t0) / 5, info


default_panel = _capi.get_option("linear_panel_features")
Xw, yw = make_blobs_pm1(20000, 64, seed=0, dtype=np.float64)
iteration_ms(Parameter(kernel_type="linear"), Xw, yw)  # (first launches of a process: module load, clocks)
for N, d in ((40000, 64), (40000, 128), (40000, 256), (40000, 320), (40000, 512), (20000, 2000)):
    X, y = make_blobs_pm1(N, d, seed=1, dtype=np.float64)
    for pf in (128, 64):
        _capi.set_option("linear_panel_features", pf)
        ms, info = iteration_ms(Parameter(kernel_type="linear"), X, y)
        print(f"fp64 {N}x{d} linear, feature panels of {pf:3d}: {ms:8.2f} ms per iteration, {2.0 * N * N * d / ms / 1e9:6.1f} TFLOP/s effective, symmetric {info.get('symmetric')}", flush=True)
    _capi.set_option("linear_panel_features", default_panel)
    if d >= 256:
        for kernel in ("polynomial", "rbf"):
            ms, info = iteration_ms(Parameter(kernel_type=kernel, gamma=1.0 / d, degree=3, coef0=1.0), X, y)
            print(f"fp64 {N}x{d} {kernel:10s}                        : {ms:8.2f} ms per iteration, {2.0 * N * N * d / ms / 1e9:6.1f} TFLOP/s effective, symmetric {info.get('symmetric')}", flush=True)
