"""GPU (-m gpu), OPT-IN (PLSSVM_AMD_HUGE=1): data sets of MORE THAN 2^31 ELEMENTS -- the "maximum sizes" edge of the hot path.

Every 32-bit quantity of the device code is a candidate for wrapping there: an element index `row * ldx` held in an int, a byte offset into the 16-bit
operand planes beyond 4 GiB, the triangular record index of the column slabs, the offsets of a rank's work items.  The reference indexes with std::size_t
throughout (include/plssvm/backends/HIP/svm_kernel.hip.hpp:40-75, src/plssvm/backends/OpenMP/svm_kernel.cpp:33-54), so it has no such edge; this library
must not have one either.  Checked the way the full-size BASELINE tests are: sampled rows of ONE implicit matvec and of the q vector against a float64
evaluation of the reference's formulas (kernel_function_types.hpp:60-97, csvm.cpp:283-306 for the rank-1 terms), the rows chosen where the offsets are
largest (the last rows of the data; for the many-point cases the LAST rank's share of a row-block sharded problem with the exchange switched off, whose own
rows are complete -- every tile at or below them belongs to it -- at 1 / world of the work).

Not part of the default run: each case needs 9-19 GB of host memory for the data set alone, which the driver's box is not known to have, and about two minutes.
Run by hand (`PLSSVM_AMD_HUGE=1 python -m pytest tests/test_gpu_huge.py -m gpu -s`); the printed table of the round's run is kept under profiles/.
"""

import os
import time

import numpy as np
import pytest

from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter
from plssvm_amd.sharding import TILE, sym_block_partition

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("PLSSVM_AMD_HUGE", "0") != "1", reason="opt-in (PLSSVM_AMD_HUGE=1): > 2^31 data elements, 9-19 GB of host memory per case")]


def _kernel_rows(kernel, X, n, rows, degree, gamma, coef0):
    """float64 rows K[rows, :n] and K[rows, N-1] (for q), evaluated chunk by chunk (the float64 copy of the whole data set would not fit)"""
    d = X.shape[1]
    chunk = max(1, (1 << 25) // d)
    xr = X[rows].astype(np.float64)
    out = np.empty((len(rows), n + 1))
    for c0 in range(0, n + 1, chunk):
        c1 = min(c0 + chunk, n + 1)
        Xc = X[c0:c1].astype(np.float64)
        G = xr @ Xc.T
        if kernel == "linear":
            out[:, c0:c1] = G
        elif kernel == "polynomial":
            out[:, c0:c1] = (gamma * G + coef0) ** degree
        else:
            sq = np.einsum("ij,ij->i", Xc, Xc)
            sr = np.einsum("ij,ij->i", xr, xr)
            out[:, c0:c1] = np.exp(-gamma * np.maximum(sr[:, None] + sq[None, :] - 2.0 * G, 0.0))
    return out


CASES = [
    # tag, kernel, dtype, points, features, world (the LAST rank's share is evaluated), what it crosses
    ("many points, 256-row workgroups", "rbf", np.float32, 8_500_001, 256, 128),
    ("many points, linear", "linear", np.float32, 8_500_001, 256, 128),
    ("wide rbf (feature panels inside a tile)", "rbf", np.float32, 600_001, 4096, 8),
    ("very wide linear (panel passes)", "linear", np.float32, 70_001, 32_768, 1),
    ("wide polynomial fp64", "polynomial", np.float64, 135_001, 16_384, 2),
]


@pytest.mark.parametrize("tag, kernel, dt, N, d, world", CASES, ids=[c[0].split(",")[0].replace(" ", "_") + "_" + c[1] for c in CASES])
def test_more_than_2_to_the_31_data_elements(tag, kernel, dt, N, d, world):
    assert N * d > 2**31
    t0 = time.time()
    X, _ = make_blobs_pm1(N, d, seed=7, dtype=dt)
    t_gen = time.time() - t0
    n = N - 1
    cost, degree, coef0, gamma = 1.0, 2, 1.0, 1.0 / d
    p = Parameter(kernel_type=kernel, degree=degree, gamma=gamma, coef0=coef0, cost=cost)
    b0, _ = sym_block_partition(n, world)[world - 1]
    r0 = b0 * TILE
    rng = np.random.default_rng(11)
    rows = np.unique(np.concatenate([[r0, r0 + 1, n - 1, n - 2], rng.integers(r0, n, size=4)]))
    v = rng.uniform(-1, 1, size=n).astype(dt)
    _capi.set_option("skip_collective", 1)
    try:
        t0 = time.time()
        with backend.ResidentProblem(p, X, rank=world - 1, world=world) as prob:
            info = prob.info()
            q, QA = prob.q()
            t_setup = time.time() - t0
            t0 = time.time()
            got = prob.matvec(v, np.zeros(n, dt), 1.0)
            t_mv = time.time() - t0
    finally:
        _capi.set_option("skip_collective", 0)
    t0 = time.time()
    K = _kernel_rows(kernel, X, n, rows, degree, gamma, coef0)
    # q_j = k(x_j, x_last); QA_cost = k(x_last, x_last) + 1 / C   (csvm.cpp:227-232, q_kernel.cpp:20-60): EVERY entry, on the scale of the vector (the linear kernel's entries cancel)
    Kq = _kernel_rows(kernel, X, n, np.array([n]), degree, gamma, coef0)[0]  # row of the last point: k(x_last, x_j) for all j, and k(x_last, x_last)
    eps = np.finfo(dt).eps
    # (k_q evaluates the reference's own fma chain over the features in the real type, q_kernel.cpp:28-33: its rounding grows with sqrt(features) -- 97 eps measured at 32 768)
    q_err = float(np.max(np.abs(q.astype(np.float64) - Kq[:n])) / np.max(np.abs(Kq[:n])))
    assert q_err <= max(64.0, 2.0 * np.sqrt(d)) * eps, q_err / eps
    assert abs(QA - (Kq[n] + 1.0 / cost)) <= 64 * eps * abs(Kq[n] + 1.0 / cost)
    v64, q64 = v.astype(np.float64), Kq[:n]
    S = v64.sum()
    want = K[:, :n] @ v64 + v64[rows] / cost + (float(QA) * S - q64 @ v64) - S * q64[rows]   # (Abar v)_i, csvm.cpp:283-306
    absv = np.abs(v64)
    scale = np.abs(K[:, :n]) @ absv + (abs(float(QA)) + np.abs(q64[rows])) * absv.sum() + np.abs(q64) @ absv + absv[rows]
    err = float(np.max(np.abs(got[rows] - want) / scale))
    err_K = float(np.max(np.abs(got[rows] - want) / (np.abs(K[:, :n]) @ absv)))  # on the scale of the kernel sum alone (the rank-1 terms are 1e3 ... 1e4 times larger here)
    t_ref = time.time() - t0
    print(f"\n[huge] {tag}: {N} x {d} {kernel} {np.dtype(dt).name}, {N * d / 2**31:.3f} x 2^31 elements, rank {world - 1} of {world} (rows {r0} ... {n - 1}), "
          f"gram_mode {info['gram_mode']}, symmetric {info['symmetric']}: q {q_err / eps:.1f} eps of its largest entry, sampled rows {err / eps:.3f} eps of the summand scale = {err_K / eps:.2f} eps of sum |K_ij v_j|, max |got| {np.max(np.abs(got[rows])):.4g}; "
          f"generate {t_gen:.0f} s, set-up {t_setup:.1f} s, matvec {t_mv:.2f} s, float64 rows {t_ref:.0f} s", flush=True)
    assert np.all(np.isfinite(got))
    assert err < 1 * eps and err_K < 64 * eps, (err / eps, err_K / eps)
