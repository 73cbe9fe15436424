#!/usr/bin/env python3
"""Randomised cross-check of the panels-inside-a-tile kernels (fp32 rbf / polynomial on wide data) against the generic native kernel:
shapes, feature counts around the one-pass limits, chunk lengths, band sizes, shard counts, plane kinds, both variants.
usage: wide_stress.py [cases] [seed] [f64]   (f64: the fp64 panel paths -- linear passes, rbf / polynomial panels inside a sub-tile)"""
import sys

import numpy as np

from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
F64 = len(sys.argv) > 3 and sys.argv[3] == "f64"
dtype = np.float64 if F64 else np.float32
names = ("gram_mode", "tile_kernel", "j_chunk_tiles", "symmetric", "colslab_band_mb", "item_order", "rbf_fold")
defaults = {k: _capi.get_option(k) for k in names}
worst = 0.0
for case in range(cases):
    kernel = ("rbf", "polynomial", "linear")[int(rng.integers(3 if F64 else 2))]
    N = int(rng.choice([2, 100, 129, 257, 640, 1000, 1537, 2500, 4100]))
    d = int(rng.choice([257, 300, 320, 449, 512, 577, 1025, 2049] if F64 else [385, 449, 512, 513, 577, 640, 700, 1025, 1500, 2049]))
    if not F64 and kernel == "polynomial" and d <= 512 and rng.integers(2):
        d += 256
    opts = dict(gram_mode=int(rng.choice([3, 1])), j_chunk_tiles=int(rng.choice([0, 1, 2, 3, 7])), symmetric=int(rng.choice([1, 1, 0])),
                colslab_band_mb=int(rng.choice([2048, 1])), item_order=int(rng.choice([0, 1, 2])), rbf_fold=int(rng.choice([1, 0])))
    shards = int(rng.choice([1, 1, 2, 3, 8]))
    degree = int(rng.choice([1, 2, 3, 4]))
    X, y = make_blobs_pm1(N, d, seed=100 + case, dtype=dtype)
    p = Parameter(kernel_type=kernel, gamma=float(rng.choice([1.0, 0.3])) / d, degree=degree, coef0=float(rng.choice([0.0, 1.0])), cost=1.0)
    v = rng.uniform(-1, 1, N - 1).astype(dtype)
    zero = np.zeros(N - 1, dtype)
    out = {}
    for label, extra in (("panels", {}), ("generic", {"tile_kernel": 1})):
        for k, val in defaults.items():
            _capi.set_option(k, val)
        for k, val in {**opts, **extra}.items():
            _capi.set_option(k, val)
        with backend.ResidentProblem(p, X, devices=[0] * shards) as prob:
            out[label] = prob.matvec(v, zero, 1.0).astype(np.float64)
            info = prob.info()
        out[label + "_mode"] = info["gram_mode"]
    # float64 truth of Abar v = K v + v / C + (QA_cost S - q.v) 1 - S q  (the yardstick: what the generic native kernel itself is off by)
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
    err_p = float(np.max(np.abs(out["panels"] - truth) / scale)) / eps
    err_g = float(np.max(np.abs(out["generic"] - truth) / scale)) / eps
    worst = max(worst, err_p)
    ok = np.all(np.isfinite(out["panels"])) and err_p < max(4.0 * err_g, 256.0 if F64 else 16.0)  # (fp64: the data carries sqrt(gamma) / the exponent scale, a power amplifies its rounding; 256 eps = 6e-14)
    print(f"case {case:3d}: {kernel:10s} N {N:5d} d {d:5d} degree {degree} coef0 {p.coef0} shards {shards} {opts} -> plane mode {out['panels_mode']}: {err_p:7.2f} eps from float64"
          f" (generic kernel: {err_g:7.2f}){'' if ok else '   <-- CHECK'}", flush=True)
for k, val in defaults.items():
    _capi.set_option(k, val)
print(f"worst: {worst:.2f} eps of the row's summands from the float64 product")
