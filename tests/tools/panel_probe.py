#!/usr/bin/env python3
"""fp32 linear kernel: features per pass of the f16x3 kernels (option linear_panel_features) at several shapes; ms per CG iteration."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plssvm_amd import _capi, backend
from plssvm_amd.parameter import Parameter
from plssvm_amd.datagen import make_blobs_pm1

for N, d in ((100000, 192), (100000, 320), (100000, 384), (100000, 512), (40000, 2000)):
    X, y = make_blobs_pm1(N, d, seed=1, dtype=np.float32)
    line = f"{N}x{d} linear:"
    for pf in (512, 256, 128, 64):
        _capi.set_option("linear_panel_features", pf)
        with backend.ResidentProblem(Parameter(kernel_type="linear"), X) as prob:
            prob.cg_begin(y, 1e-30); prob.cg_step(2); prob.synchronize()
            t0 = time.perf_counter(); prob.cg_step(6); prob.synchronize(); t1 = time.perf_counter()
            line += f"  panel {pf}: {(t1 - t0) / 6 * 1e3:7.2f} ms"
    print(line, flush=True)
