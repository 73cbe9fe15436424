#!/usr/bin/env python3
"""Developer probe (not a test): data far wider than any BASELINE configuration -- the reference's scaling studies go to 2^14 features -- through
the public solve path: one implicit matvec against a float64 numpy product, all three kernels, both real types.
usage: very_wide_probe.py [N d]..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

shapes = [(int(a), int(b)) for a, b in zip(sys.argv[1::2], sys.argv[2::2])] or [(3000, 16384), (1500, 65536)]
for N, d in shapes:
    for dtype in (np.float32, np.float64):
        X, _ = make_blobs_pm1(N, d, seed=5, dtype=dtype)
        v = np.random.default_rng(2).uniform(-1, 1, N - 1).astype(dtype)
        Xa = X.astype(np.float64)
        G = Xa @ Xa.T
        sq = np.einsum("ij,ij->i", Xa, Xa)
        for kernel in ("linear", "polynomial", "rbf"):
            p = Parameter(kernel_type=kernel, gamma=1.0 / d, degree=3, coef0=0.5, cost=1.0)
            K = G if kernel == "linear" else ((G / d + 0.5) ** 3 if kernel == "polynomial" else np.exp(-(sq[:, None] + sq[None, :] - 2 * G) / d))
            n = N - 1
            Kn, q, QA = K[:n, :n], K[:n, n], K[n, n] + 1.0
            v64 = v.astype(np.float64)
            S = v64.sum()
            truth = Kn @ v64 + v64 + (QA * S - q @ v64) - S * q
            scale = np.abs(Kn) @ np.abs(v64) + np.abs(v64) + abs(QA * S) + abs(q @ v64) + np.abs(S * q)
            t0 = time.time()
            with backend.ResidentProblem(p, X) as prob:
                out = prob.matvec(v, np.zeros(n, dtype), 1.0).astype(np.float64)
                info = prob.info()
            err = float(np.max(np.abs(out - truth) / scale)) / float(np.finfo(dtype).eps)
            print(f"{N} x {d} {np.dtype(dtype).name} {kernel:10s}: {err:8.2f} eps of the row's summands from float64; gram_mode {info['gram_mode']} symmetric {info['symmetric']} "
                  f"tile kernels {info['matvec_kernel_ms']:.2f} ms ({time.time() - t0:.1f} s with set-up)", flush=True)
