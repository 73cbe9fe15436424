#!/usr/bin/env python3
"""One-off parity measurement at a size the CPU reference still finishes in minutes (run on the GPU box):
fp32 CG to eps on N x d rbf / linear data -- GPU default (f16x3), GPU bf16x6, GPU native v_mfma_f32, the reference's OpenMP kernels in fp32 --
each measured against the GPU fp64 solve of the same system.  usage: parity_at_scale.py [N] [d] [eps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402
import oracle_lib  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
eps = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
max_iter = 400


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / np.max(np.abs(np.asarray(b, np.float64))))


for kernel in ("rbf", "linear"):
    X32, y32 = make_blobs_pm1(N, d, seed=5, dtype=np.float32)
    p = Parameter(kernel_type=kernel, cost=1.0)
    t0 = time.time()
    a64, r64, i64 = backend.solve_system_of_linear_equations(p, X32.astype(np.float64), y32.astype(np.float64), eps, max_iter)
    print(f"{kernel} {N}x{d} eps={eps}: GPU fp64 {i64['iterations']} its, {time.time() - t0:.1f} s", flush=True)
    for mode, name in ((3, "GPU fp32 f16x3 (default)"), (1, "GPU fp32 bf16x6"), (0, "GPU fp32 native v_mfma_f32")):
        _capi.set_option("gram_mode", mode)
        a, r, info = backend.solve_system_of_linear_equations(p, X32, y32, eps, max_iter)
        print(f"  {name:30s} its {info['iterations']:4d}  alpha rel-inf vs fp64 {rel_inf(a, a64):.3e}  rho {abs(r - r64) / abs(r64):.3e}", flush=True)
    _capi.set_option("gram_mode", 3)
    impl = oracle_lib.ref() if oracle_lib.have_ref() else oracle_lib.oracle()
    t0 = time.time()
    a, r, info = impl.solve(kernel, X32, y32, eps, max_iter, gamma=1.0 / d, degree=3, coef0=0.0, cost=1.0)
    print(f"  {'reference OpenMP fp32':30s} its {int(info['iterations']):4d}  alpha rel-inf vs fp64 {rel_inf(a, a64):.3e}  rho {abs(r - r64) / abs(r64):.3e}   ({time.time() - t0:.0f} s, "
          f"{'reference kernels' if oracle_lib.have_ref() else 'oracle port'}, {impl.num_threads() if hasattr(impl, 'num_threads') else '?'} threads)", flush=True)
