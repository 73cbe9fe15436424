#!/usr/bin/env python3
"""Diagnostic: the software-pipelined kernel (mfma_shape 3) against the two-waves-per-SIMD kernel (mfma_shape 2): bit-identical?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plssvm_amd import _capi, backend
from plssvm_amd.parameter import Parameter
from plssvm_amd.datagen import make_blobs_pm1

d = 128
for N, jct in ((129, 0), (257, 0), (385, 0), (1500, 0), (1500, 1), (1500, 3), (1500, 5), (4097, 0), (20000, 0)):
    X, y = make_blobs_pm1(N, d, seed=3, dtype=np.float32)
    n = N - 1
    v = np.random.default_rng(1).uniform(-1, 1, n).astype(np.float32)
    out = {}
    for shape in (2, 3):
        for k, val in (("gram_mode", 2), ("mfma_shape", shape), ("symmetric", 1), ("rbf_fold", 1), ("j_chunk_tiles", jct)):
            _capi.set_option(k, val)
        with backend.ResidentProblem(Parameter(kernel_type="rbf", gamma=1.0 / d), X) as prob:
            out[shape] = prob.matvec(v, np.zeros(n, np.float32), 1.0)
    bad = out[2] != out[3]
    nan = ~np.isfinite(out[3])
    print(f"N={N} jc_tiles={jct}: equal {bool(np.array_equal(out[2], out[3]))} differing {int(bad.sum())} nan {int(nan.sum())}"
          + (f" first {int(np.argmax(bad))} last {int(len(bad) - 1 - np.argmax(bad[::-1]))} max rel {float(np.nanmax(np.abs(out[2] - out[3]) / (np.abs(out[2]) + 1e-30))):.3g}" if bad.any() else ""), flush=True)
