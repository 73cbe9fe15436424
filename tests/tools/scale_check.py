#!/usr/bin/env python3
"""Scale check on one MI355X beyond BASELINE's largest configuration: N x 128 rbf fp32 (default N = 2 000 000: 62.5 GB of column
records), two CG iterations for the timing and 64 sampled rows of one implicit matvec against a float64 numpy evaluation.
usage: scale_check.py [N]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

N, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 2000000), 128
X, y = make_blobs_pm1(N, d, seed=1, dtype=np.float32)
p = Parameter(kernel_type="rbf")
t = time.time()
with backend.ResidentProblem(p, X) as prob:
    print(f"setup {time.time() - t:.1f} s", flush=True)
    prob.cg_begin(y, 1e-30)
    prob.cg_step(2)
    prob.synchronize()
    i = prob.info()
    print(f"{N} x {d}: tile kernel {i['matvec_kernel_ms']:.1f} ms, symmetric {i['symmetric']}, gram_mode {i['gram_mode']}", flush=True)
    v = np.random.default_rng(0).uniform(-1, 1, N - 1).astype(np.float32)
    out = prob.matvec(v, np.zeros(N - 1, np.float32), 1.0)
    q, QA = prob.q()
rows = np.random.default_rng(1).choice(N - 1, 64, replace=False)
X64, v64, q64 = X.astype(np.float64), v.astype(np.float64), q.astype(np.float64)
S, qv = v64.sum(), q64 @ v64
want, scale = [], []
for r in rows:
    k = np.exp(-((X64[: N - 1] - X64[r]) ** 2).sum(1) / d)
    want.append(k @ v64 + v64[r] + float(QA) * S - qv - S * q64[r])
    scale.append(k @ np.abs(v64) + abs(v64[r]) + abs(float(QA) * S) + abs(qv) + abs(S * q64[r]))
want, scale = np.array(want), np.array(scale)
print(f"64 sampled rows of A-bar v against float64: max rel err {np.max(np.abs(out[rows] - want)) / np.max(np.abs(want)):.3e}; "
      f"worst row {np.max(np.abs(out[rows] - want) / scale) / 2.0 ** -24:.2f} eps of the row's summands", flush=True)
