#!/usr/bin/env python3
"""Diagnostic: where are the wrong rows?  rbf, f16x3, hand-scheduled, symmetric, folded records; several sizes and chunk lengths."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plssvm_amd import _capi, backend
from plssvm_amd.parameter import Parameter
from plssvm_amd.datagen import make_blobs_pm1

def runs(mask):
    out, i, n = [], 0, len(mask)
    while i < n:
        if mask[i]:
            j = i
            while j + 1 < n and mask[j + 1]: j += 1
            out.append((i, j)); i = j + 1
        else: i += 1
    return out

d = 128
for N in (257, 385, 641, 1500):
    X, y = make_blobs_pm1(N, d, seed=3, dtype=np.float32)
    X64 = X.astype(np.float64); n = N - 1
    sq = np.einsum("ij,ij->i", X64[:n], X64[:n])
    K = np.exp(-(1.0 / d) * np.maximum(sq[:, None] + sq[None, :] - 2 * X64[:n] @ X64[:n].T, 0))
    for vname, v in (("ones", np.ones(n, np.float32)), ("e0", np.eye(n, dtype=np.float32)[0]), ("e200", np.eye(n, dtype=np.float32)[min(200, n - 1)])):
        for jct in (1, 2, 4):
            for k, val in (("gram_mode", 2), ("mfma_shape", 2), ("symmetric", 1), ("rbf_fold", 1), ("j_chunk_tiles", jct)):
                _capi.set_option(k, val)
            with backend.ResidentProblem(Parameter(kernel_type="rbf", gamma=1.0 / d), X) as prob:
                q, QA = prob.q()
                got = prob.matvec(v, np.zeros(n, np.float32), 1.0).astype(np.float64)
            S = float(v.sum()); qv = float(q.astype(np.float64) @ v)
            want = K @ v + v + (QA * S - qv) - S * q.astype(np.float64)
            bad = ~np.isfinite(got) | (np.abs(got - want) > 1e-3 * (np.abs(K) @ np.abs(v) + abs(QA * S) + np.abs(S * q) + 1e-30))
            print(f"N={N} v={vname:5s} jc_tiles={jct}: wrong rows {runs(bad)[:8]} nan {int((~np.isfinite(got)).sum())}", flush=True)
