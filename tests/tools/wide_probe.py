#!/usr/bin/env python3
"""Timing of one implicit matvec on wide fp32 rbf / polynomial data: the split kernels over feature panels inside a tile (default) against the
generic native kernel (option tile_kernel = 1, full square).  usage: wide_probe.py [N d]..."""
import sys
import time

import numpy as np

from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter

shapes = [(40000, 2000), (60000, 640), (100000, 512 + 128)]
if len(sys.argv) > 2:
    shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
defaults = {name: _capi.get_option(name) for name in ("gram_mode", "tile_kernel")}
for N, d in shapes:
    X, y = make_blobs_pm1(N, d, seed=1, dtype=np.float32)
    for kernel in ("rbf", "polynomial"):
        for label, opts in (("split kernels, feature panels in the tile", {}), ("same on bf16x6 planes", {"gram_mode": 1}), ("generic native kernel (full square)", {"tile_kernel": 1})):
            for k, v in defaults.items():
                _capi.set_option(k, v)
            for k, v in opts.items():
                _capi.set_option(k, v)
            p = Parameter(kernel_type=kernel, gamma=1.0 / d, degree=3, coef0=1.0, cost=1.0)
            with backend.ResidentProblem(p, X) as prob:
                prob.cg_begin(y, 1e-30)
                prob.cg_step(2)
                t0 = time.perf_counter()
                prob.cg_step(5)
                info = prob.cg_finish()[2]
                dt = (time.perf_counter() - t0) / 5
            print(f"{N}x{d} {kernel:10s} {label:45s}: {1e3 * dt:8.2f} ms per CG iteration, {2.0 * N * N * d / dt / 1e12:7.1f} TFLOP/s effective; gram_mode {info.get('gram_mode')} symmetric {info.get('symmetric')}", flush=True)
for k, v in defaults.items():
    _capi.set_option(k, v)
