#!/usr/bin/env python3
"""Developer check of the 256-row-workgroup kernels (mfma_shape 3) against the 128-row kernels (mfma_shape 2) and a float64 numpy product:
work-item shapes from one tile to many, odd and even numbers of row blocks, several chunk lengths, bands, shards, every lag the build has."""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402


def kmat(kernel, X, gamma, degree=3, coef0=0.5):
    X = X.astype(np.float64)
    G = X @ X.T
    if kernel == "linear":
        return G
    if kernel == "polynomial":
        return (gamma * G + coef0) ** degree
    sq = np.diag(G)
    return np.exp(-gamma * (sq[:, None] + sq[None, :] - 2 * G))


def main():
    lags = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1").split(",")]
    defaults = {n: _capi.get_option(n) for n in _capi.OPTION_NAMES}
    worst = 0.0
    nbad = 0
    for N, d, kernel, jct, band in itertools.product((130, 258, 386, 1500, 3000, 4226), (128, 100), ("rbf", "linear", "polynomial"), (0, 1, 5), (2048, 1)):
        if band == 1 and N < 3000:
            continue
        X, _ = make_blobs_pm1(N, d, seed=N + d, dtype=np.float32)
        v = np.random.default_rng(N).uniform(-1, 1, N - 1).astype(np.float32)
        p = Parameter(kernel_type=kernel, gamma=1.0 / d, degree=3, coef0=0.5)
        K = kmat(kernel, X[:-1], 1.0 / d)
        ref = K @ v.astype(np.float64)
        scale = np.abs(K) @ np.abs(v.astype(np.float64))
        out = {}
        for shape, lag in [(2, 0)] + [(3, l) for l in lags]:
            for n, val in defaults.items():
                _capi.set_option(n, val)
            for k, val in (("gram_mode", 2), ("mfma_shape", shape), ("pair_lag", lag), ("j_chunk_tiles", jct), ("colslab_band_mb", band)):
                _capi.set_option(k, val)
            with backend.ResidentProblem(p, X) as prob:
                q, QA = prob.q()
                got = prob.matvec(v, np.zeros(N - 1, np.float32), 1.0).astype(np.float64)
                # remove the rank-1 terms: Abar v = K v + v / C + (QA S - q.v) 1 - S q
                S, qv = float(np.sum(v.astype(np.float64))), float(np.dot(q.astype(np.float64), v.astype(np.float64)))
                kv = got - v / 1.0 - (QA * S - qv) + S * q.astype(np.float64)
                out[(shape, lag)] = kv
        for key, kv in out.items():
            err = float(np.max(np.abs(kv - ref) / scale)) / 2.0 ** -24
            worst = max(worst, err)
            flag = "" if err < 16 and np.all(np.isfinite(kv)) else "   <-- BAD"
            nbad += 1 if flag else 0
            print(f"N {N:5d} d {d:3d} {kernel:10s} jct {jct} band {band:4d} shape {key[0]} lag {key[1]}: max err {err:6.2f} eps of the row's summands{flag}", flush=True)
    print(f"worst {worst:.2f} eps, {nbad} bad")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main())
