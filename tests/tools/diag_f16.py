#!/usr/bin/env python3
"""Diagnostic: one implicit matvec per option combination against a float64 numpy evaluation (small sizes); prints error and NaN positions."""
import itertools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plssvm_amd import _capi, backend
from plssvm_amd.parameter import Parameter
from plssvm_amd.datagen import make_blobs_pm1

def truth(kernel, X, v, gamma):
    X64 = X.astype(np.float64)
    n = X.shape[0] - 1
    G = X64[:n] @ X64[:n].T
    if kernel == "rbf":
        sq = np.einsum("ij,ij->i", X64[:n], X64[:n])
        K = np.exp(-gamma * np.maximum(sq[:, None] + sq[None, :] - 2 * G, 0))
    elif kernel == "linear":
        K = G
    else:
        K = (gamma * G) ** 3
    return K @ v.astype(np.float64), np.abs(K) @ np.abs(v.astype(np.float64))

cases = [(1500, 96), (1500, 128), (700, 64)] if len(sys.argv) < 2 else [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
for (N, d) in cases:
    X, y = make_blobs_pm1(N, d, seed=3, dtype=np.float32)
    v = np.random.default_rng(1).uniform(-1, 1, N - 1).astype(np.float32)
    for kernel in ("rbf", "linear"):
        Kv, scale = truth(kernel, X, v, 1.0 / d)
        for gm, shape, sym, fold in itertools.product((2, 1), (3, 2), (1, 0), (1, 0)):
            if kernel != "rbf" and fold == 0:
                continue
            for k, val in (("gram_mode", gm), ("mfma_shape", shape), ("symmetric", sym), ("rbf_fold", fold)):
                _capi.set_option(k, val)
            p = Parameter(kernel_type=kernel, gamma=1.0 / d)
            with backend.ResidentProblem(p, X) as prob:
                q, QA = prob.q()
                # A v = K v + v / C + (QA S - q.v) 1 - S q
                got = prob.matvec(v, np.zeros(N - 1, np.float32), 1.0).astype(np.float64)
                info = prob.info()
            S = float(v.astype(np.float64).sum()); qv = float(q.astype(np.float64) @ v.astype(np.float64))
            want = Kv + v + (QA * S - qv) - S * q.astype(np.float64)
            bad = ~np.isfinite(got)
            err = np.abs(got - want) / (scale + abs(QA * S) + np.abs(S * q))
            print(f"{N}x{d} {kernel:6s} gm{gm}->{info['gram_mode']} shape{shape} sym{sym} fold{fold}: nan {int(bad.sum()):5d}"
                  + (f" first {int(np.argmax(bad))} last {int(len(bad) - 1 - np.argmax(bad[::-1]))}" if bad.any() else "")
                  + f"  max err/eps {np.nanmax(err) / np.finfo(np.float32).eps:8.2f}", flush=True)
