#!/usr/bin/env python3
"""Where a SHORT solve's time goes (the reference's default epsilon gives two iterations on the blobs of BASELINE configs[1]): problem set-up, cg_begin, the iterations,
cg_finish, tear-down -- through the resident-problem entry points, and the one-shot lssvm_mi355_solve_* call beside them.
usage: solve_phases.py [num_points [num_features]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from plssvm_amd import backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
X, y = make_blobs_pm1(n, d, seed=42, dtype=np.float32)
p = Parameter(kernel_type="rbf")
for rep in range(4):
    t0 = time.perf_counter()
    prob = backend.ResidentProblem(p, X)
    t1 = time.perf_counter()
    prob.cg_begin(y, 1e-3)
    t2 = time.perf_counter()
    done = prob.cg_step(2)
    t3 = time.perf_counter()
    alpha, rho, info = prob.cg_finish()
    t4 = time.perf_counter()
    prob.close()
    t5 = time.perf_counter()
    print(f"resident: create {1e3 * (t1 - t0):.2f} ms (set-up {info['setup_ms']:.2f}), cg_begin {1e3 * (t2 - t1):.2f}, 2 iterations {1e3 * (t3 - t2):.2f}, cg_finish {1e3 * (t4 - t3):.2f}, "
          f"close {1e3 * (t5 - t4):.2f}  -> {1e3 * (t5 - t0):.2f} ms", flush=True)
for rep in range(4):
    t0 = time.perf_counter()
    alpha, rho, info = backend.solve_system_of_linear_equations(p, X, y, 1e-3, n)
    t1 = time.perf_counter()
    print(f"one-shot solve: {1e3 * (t1 - t0):.2f} ms (set-up {info['setup_ms']:.2f}, cg {info['total_ms']:.2f}, {info['iterations']} iterations)", flush=True)
