#!/usr/bin/env python3
"""Diagnostic for the wrong instantiation of tile_matvec_f64_wide<KT_POLY, true> (run-time integer power, symmetric variant) when its two outer loops
are peeled by the optimiser (lssvm_tile_f64_wide.hip.hpp): which rows are off, by what, per wave and lane group.
usage: PLSSVM_AMD_LIBRARY=<variant build> miscompile_f64_wide.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

N, d = 513, 320
X, y = make_blobs_pm1(N, d, seed=166, dtype=np.float64)
n = N - 1
worst_all = 0.0
for degree, coef0 in ((4, 1.0), (1, 0.0), (5, 0.5)):
    p = Parameter(kernel_type="polynomial", gamma=0.3 / d, degree=degree, coef0=coef0, cost=1.0)
    v = np.ones(n)
    _capi.set_option("symmetric", 1)
    with backend.ResidentProblem(p, X) as prob:
        out = prob.matvec(v, np.zeros(n), 1.0)
    Ka = (p.gamma * (X @ X.T) + coef0) ** degree
    K, q, QA = Ka[:n, :n], Ka[:n, n], Ka[n, n] + 1.0
    S = float(v.sum())
    truth = K @ v + v + (QA * S - float(q @ v)) - S * q
    scale = np.abs(K) @ np.abs(v) + np.abs(v) + abs(QA * S) + abs(float(q @ v)) + np.abs(S * q)
    rel = np.abs(out - truth) / scale / np.finfo(np.float64).eps
    bad = np.nonzero(rel > 64)[0]
    worst_all = max(worst_all, float(rel.max()))
    print(f"degree {degree} coef0 {coef0}: worst {rel.max():.3g} eps, {bad.size} rows off")
    if bad.size:
        # row = 128 block + 32 wave + 16 rb + 4 i + q  (lssvm_tile_f64_wide.hip.hpp: rowpart[rb][i] of lane group q)
        import collections
        groups = collections.defaultdict(list)
        avg = float(np.mean(np.abs(K)))
        for r in bad:
            groups[(int(r) // 128, (int(r) % 128) // 32, (int(r) % 32) // 16, (int(r) % 16) // 4)].append((int(r) % 4, round(float(out[r] - truth[r]) / avg, 1)))
        for key in sorted(groups):
            print("   (block, wave, rb, i) =", key, " (q, columns' worth of error):", groups[key])
        for r in bad[:8]:
            print(f"   row {r}: got {out[r]:.12g} want {truth[r]:.12g} diff {out[r] - truth[r]:.6g}")
print("RESULT", "WRONG" if worst_all > 64 else "right")
