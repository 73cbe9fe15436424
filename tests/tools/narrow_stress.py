#!/usr/bin/env python3
"""Randomised cross-check of the resident-row-panel tile kernels (up to 512 features: the hand-scheduled and compiler-scheduled f16x3 / bf16x6
kernels, the native fp32 kernel, the fp64 kernel, the linear kernel's panel passes) against the float64 product and the generic kernel:
shapes, feature counts, kernels, degrees, both types, Gram modes, chunk lengths, band sizes, shard counts, item orders, both variants.
usage: narrow_stress.py [cases] [seed]"""
import sys

import numpy as np

from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
names = ("gram_mode", "tile_kernel", "j_chunk_tiles", "symmetric", "colslab_band_mb", "item_order", "rbf_fold", "mfma_shape")
defaults = {k: _capi.get_option(k) for k in names}
worst = 0.0
flags = 0
for case in range(cases):
    dtype = (np.float32, np.float64)[int(rng.integers(2))]
    kernel = ("rbf", "polynomial", "linear")[int(rng.integers(3))]
    N = int(rng.choice([2, 3, 100, 128, 129, 130, 257, 640, 1000, 1537, 2500, 4100]))
    d = int(rng.choice([1, 3, 16, 31, 64, 65, 100, 128, 129, 192, 200, 256, 257, 300, 384, 385, 448, 512]))
    opts = dict(gram_mode=int(rng.choice([3, 3, 1, 0, 2])), j_chunk_tiles=int(rng.choice([0, 0, 1, 2, 3, 7])), symmetric=int(rng.choice([1, 1, 0])),
                colslab_band_mb=int(rng.choice([2048, 1])), item_order=int(rng.choice([0, 1, 2])), rbf_fold=int(rng.choice([1, 0])), mfma_shape=int(rng.choice([2, 2, 1])))
    shards = int(rng.choice([1, 1, 2, 3, 8]))
    degree = int(rng.choice([0, 1, 2, 3, 4]))
    X, y = make_blobs_pm1(N, d, seed=300 + case, dtype=dtype)
    p = Parameter(kernel_type=kernel, gamma=float(rng.choice([1.0, 0.3])) / d, degree=degree, coef0=float(rng.choice([0.0, 1.0])), cost=1.0)
    v = rng.uniform(-1, 1, N - 1).astype(dtype)
    zero = np.zeros(N - 1, dtype)
    out = {}
    for label, extra in (("tiles", {}), ("generic", {"tile_kernel": 1})):
        for k, val in defaults.items():
            _capi.set_option(k, val)
        for k, val in {**opts, **extra}.items():
            _capi.set_option(k, val)
        with backend.ResidentProblem(p, X, devices=[0] * shards) as prob:
            out[label] = prob.matvec(v, zero, 1.0).astype(np.float64)
            info = prob.info()
        out[label + "_info"] = (info["gram_mode"], info["symmetric"])
    Xa = X.astype(np.float64)
    Ga = Xa @ Xa.T
    if kernel == "linear":
        Ka = Ga
    elif kernel == "polynomial":
        Ka = (p.gamma * Ga + p.coef0) ** degree
    else:
        sq = np.einsum("ij,ij->i", Xa, Xa)
        Ka = np.exp(-p.gamma * np.maximum(sq[:, None] + sq[None, :] - 2.0 * Ga, 0.0))
    n = N - 1
    K, q, QA = Ka[:n, :n], Ka[:n, n], Ka[n, n] + 1.0
    v64 = v.astype(np.float64)
    S = float(v64.sum())
    truth = K @ v64 + v64 + (QA * S - float(q @ v64)) - S * q
    scale = np.abs(K) @ np.abs(v64) + np.abs(v64) + abs(QA * S) + abs(float(q @ v64)) + np.abs(S * q)
    eps = np.finfo(dtype).eps
    err_t = float(np.max(np.abs(out["tiles"] - truth) / scale)) / eps
    err_g = float(np.max(np.abs(out["generic"] - truth) / scale)) / eps
    worst = max(worst, err_t)
    ok = np.all(np.isfinite(out["tiles"])) and err_t < max(4.0 * err_g, 256.0 if dtype == np.float64 else 16.0)
    flags += 0 if ok else 1
    print(f"case {case:3d}: {np.dtype(dtype).name} {kernel:10s} N {N:5d} d {d:4d} degree {degree} coef0 {p.coef0} shards {shards} {opts} -> (gram mode, symmetric) {out['tiles_info']}: {err_t:7.2f} eps from float64"
          f" (generic kernel: {err_g:7.2f}){'' if ok else '   <-- CHECK'}", flush=True)
for k, val in defaults.items():
    _capi.set_option(k, val)
print(f"worst: {worst:.2f} eps of the row's summands from the float64 product; {flags} case(s) flagged")
