#!/usr/bin/env python3
"""Developer probe: is one implicit matvec bitwise reproducible over fresh problems of one process?  usage: determinism_probe.py kernel dtype N d [repeats]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import backend  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

kernel, dtype, N, d = sys.argv[1], np.dtype(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 30
rng = np.random.default_rng(2000)
X = rng.uniform(-1, 1, size=(N, d)).astype(dtype)
rhs = rng.uniform(-1, 1, size=N - 1).astype(dtype)
zero = np.zeros(N - 1, dtype)
p = Parameter(kernel_type=kernel, degree=2, gamma=0.001, coef0=1.0, cost=0.1)
seen = {}
first = None
for i in range(reps):
    with backend.ResidentProblem(p, X) as prob:
        out = prob.matvec(rhs, zero, 1.0)
        out2 = prob.matvec(rhs, zero, 1.0)
    h = hashlib.sha256(out.tobytes()).hexdigest()[:12]
    h2 = hashlib.sha256(out2.tobytes()).hexdigest()[:12]
    if first is None:
        first = out.copy()
    seen[h] = seen.get(h, 0) + 1
    seen[h2] = seen.get(h2, 0) + 1
    if h != h2 or not np.array_equal(out, first):
        diff = np.abs(out.astype(np.float64) - first.astype(np.float64))
        print(f"  instance {i}: first matvec {h}, second {h2}; differs from instance 0 in {int((diff > 0).sum())} entries, max {diff.max():.3e} at {int(diff.argmax())}")
print(f"{kernel} {dtype.name} {N} x {d}: {len(seen)} distinct result(s) over {2 * reps} matvecs: {seen}")
