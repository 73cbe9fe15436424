#!/usr/bin/env python3
"""Diagnostic 2 for the wrong instantiation of tile_matvec_f64_wide<KT_POLY, true>: two row blocks, v = unit vector of a row of block 1, so that rows 0..127 of
K v are exactly the MIRRORED column sums of tile (1, 0) = K[k, 0:128]: which columns are wrong, and what do they hold instead?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

N, d = 257, 320
X, y = make_blobs_pm1(N, d, seed=166, dtype=np.float64)
n = N - 1
p = Parameter(kernel_type="polynomial", gamma=0.3 / d, degree=4, coef0=1.0, cost=1.0)
Ka = (p.gamma * (X @ X.T) + 1.0) ** 4
K, q, QA = Ka[:n, :n], Ka[:n, n], Ka[n, n] + 1.0
_capi.set_option("symmetric", 1)
with backend.ResidentProblem(p, X) as prob:
    for k in (128, 133, 144, 159, 160, 175, 188, 228, 255):
        v = np.zeros(n)
        v[k] = 1.0
        out = prob.matvec(v, np.zeros(n), 1.0)
        S = 1.0
        kv = out - v - (QA * S - q[k]) + S * q      # = K v
        want = K[:, k]
        err = np.abs(kv - want) / np.abs(want)
        bad = np.nonzero(err > 1e-10)[0]
        badc = [int(b) for b in bad if b < 128]
        badr = [int(b) for b in bad if b >= 128]
        print(f"unit row {k} (wave {(k % 128) // 32} rb {(k % 32) // 16} i {(k % 16) // 4} q {k % 4}): wrong mirrored columns {badc}  wrong rows of block 1 {badr}")
        for c in badc[:6]:
            # does the wrong value equal K[k', c] for another row k' of block 1, or a sum of some?
            cand = [int(kk) for kk in range(128, 256) if abs(K[kk, c] - kv[c]) < 1e-9 * abs(kv[c]) + 1e-300]
            print(f"    column {c}: got {kv[c]:.12g} want {want[c]:.12g}; equals K[k', c] for k' in {cand}")
