#!/usr/bin/env python3
"""Developer tool (not a test): is the fp32 rho of a split Gram mode (f16x3, the default; bf16x6) systematically worse than that of the native v_mfma_f32 mode?
(16384 x 128 linear, eps 1e-6: |rho - rho64| / |rho64| = 6.9 for bf16x6, 0.31 native, 1.28 for the reference.)

For several seeds: fp32 CG to eps on N x d data with both Gram modes and -- optionally -- the reference's OpenMP kernels, each against the
GPU fp64 solve.  Prints rho64 itself, the ABSOLUTE rho errors and alpha's rel-inf error: rho = -(y_N + QA_cost sum(x) - q.x) is a difference
of two sums of n terms each, so its natural error scale is eps32 * (|QA_cost| sum|x| + sum|q x|), printed as `scale`.
usage: rho_study.py [N] [d] [eps] [seeds] [--ref]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402
import oracle_lib  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 8192
d = int(args[1]) if len(args) > 1 else 128
eps = float(args[2]) if len(args) > 2 else 1e-6
seeds = int(args[3]) if len(args) > 3 else 4
with_ref = "--ref" in sys.argv


def rel_inf(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))) / np.max(np.abs(np.asarray(b, np.float64))))


for kernel in ("linear", "rbf"):
    for seed in range(1, seeds + 1):
        X32, y32 = make_blobs_pm1(N, d, seed=seed, dtype=np.float32)
        p = Parameter(kernel_type=kernel, cost=1.0)
        X64, y64 = X32.astype(np.float64), y32.astype(np.float64)
        a64, r64, i64 = backend.solve_system_of_linear_equations(p, X64, y64, eps, 400)
        q = backend.generate_q(p, X64)
        QA = (float(X64[-1] @ X64[-1]) if kernel == "linear" else 1.0) + 1.0
        scale = np.finfo(np.float32).eps * (abs(QA) * np.abs(a64[:-1]).sum() + np.abs(q * a64[:-1]).sum())
        line = f"{kernel:6s} {N}x{d} seed {seed}: fp64 {i64['iterations']:3d} its rho64 {float(r64):+.4e}  scale {scale:.2e} |"
        for mode, name in ((3, "f16x3"), (1, "bf16x6"), (0, "native")):
            _capi.set_option("gram_mode", mode)
            a, r, info = backend.solve_system_of_linear_equations(p, X32, y32, eps, 400)
            line += f" {name}: its {info['iterations']:3d} alpha {rel_inf(a, a64):.2e} |drho| {abs(float(r) - float(r64)):.2e} |"
        _capi.set_option("gram_mode", 3)
        if with_ref and oracle_lib.have_ref():
            a, r, info = oracle_lib.ref().solve(kernel, X32, y32, eps, 400, gamma=1.0 / d, degree=3, coef0=0.0, cost=1.0)
            line += f" reference fp32: its {int(info['iterations']):3d} alpha {rel_inf(a, a64):.2e} |drho| {abs(float(r) - float(r64)):.2e}"
        print(line, flush=True)
