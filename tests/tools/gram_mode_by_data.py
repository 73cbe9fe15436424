#!/usr/bin/env python3
"""Which Gram mode does data get that is NOT min-max-scaled Gaussian blobs?  

fp32 problems run their Gram tiles on the 16-bit matrix cores: "f16x3" (two f16 planes per operand, three plane products) where the set-up MEASURES that two f16 planes
represent this data as well as fp32 does, else "bf16x6" (exact three-way bf16 split, six plane products: 1.66x the time at configs[4]) -- gram_mode 3, the default.
The reference's arithmetic is data independent (/root/reference/src/plssvm/backends/OpenMP/svm_kernel.cpp:33-54); this table shows what the check does to data of
other shapes, 50 000 x 128 each unless noted:

    blobs          two Gaussian blobs, min-max scaled to [-1, 1] per feature (the bench's data, utility_scripts/generate_data.py's recipe)
    sparse01       5 % ones, the rest zeros (a densified bag of words)
    counts         Poisson(3) word counts, unscaled small integers
    pixels         integers 0 ... 255, unscaled
    lognormal      exp(N(0, 1.5^2)) per entry, unscaled, heavy tailed (entries from 1e-3 to 1e3)
    lognormal_mm   the same, min-max scaled to [-1, 1] per feature (what plssvm-scale would hand to plssvm-train)
    wide_range     N(0, 1) x 10^U(-6, 6) per FEATURE: columns of very different magnitude, unscaled
    row_scales     N(0, 1) x 10^U(-4, 4) per POINT: rows of very different magnitude, unscaled (the one shape here that two f16 planes with ONE scale cannot hold)
    ref500x200     the reference's own tests/data/libsvm/500x200.libsvm (from the committed golden inputs; 500 x 200)

Per data set and kernel (linear; rbf with gamma = 1 / num_features): the representability statistic the library measured (lssvm_cg_info.f16_row_rel_error; accepted up
to 2^-22 = 2.4e-7 -- rbf also under an absolute bound on the exponent), the exponent scale of the rbf kernel, the mode that ran, the tile kernel's time per implicit matvec, and the error of
one implicit matvec on 48 sampled rows against the float64 oracle in units of fp32 eps of each row's summands -- with the mode FORCED to the other split as well, so that
what the check buys (or costs) is visible.  Run on the GPU box:  python tests/tools/gram_mode_by_data.py > gpurun_out/gram_mode_by_data.log
"""

import os

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # idle OpenMP workers of the oracle calls sleep instead of spinning beside the GPU legs
os.environ.setdefault("GOMP_SPINCOUNT", "0")
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from plssvm_amd import backend  # noqa: E402
from plssvm_amd._capi import Options  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

MODES = {0: "native f32", 1: "bf16x6", 2: "f16x3", 3: "f16 grid planes"}


def minmax(X):
    lo, hi = X.min(axis=0), X.max(axis=0)
    span = np.where(hi > lo, hi - lo, 1.0)
    return (-1.0 + 2.0 * (X - lo) / span).astype(np.float32)


def data_sets(n, d, seed=7):
    rng = np.random.default_rng(seed)
    y = np.where(np.arange(n) % 2 == 0, 1.0, -1.0).astype(np.float32)
    shift = (y[:, None] > 0) * 0.5
    yield "blobs", make_blobs_pm1(n, d, seed=42, dtype=np.float32)[0], y
    yield "sparse01", (rng.random((n, d)) < 0.05 + 0.03 * shift).astype(np.float32), y
    yield "counts", rng.poisson(3.0 + shift, size=(n, d)).astype(np.float32), y
    yield "pixels", rng.integers(0, 256, size=(n, d)).astype(np.float32), y
    logn = np.exp(rng.normal(0.0, 1.5, size=(n, d)) + shift)
    yield "lognormal", logn.astype(np.float32), y
    yield "lognormal_mm", minmax(logn), y
    yield "wide_range", (rng.normal(0, 1, size=(n, d)) * 10.0 ** rng.uniform(-6, 6, size=(1, d))).astype(np.float32), y
    yield "row_scales", (rng.normal(0, 1, size=(n, d)) * 10.0 ** rng.uniform(-4, 4, size=(n, 1))).astype(np.float32), y
    inputs = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
    if "500x200_X" in inputs:
        yield "ref500x200", inputs["500x200_X"].astype(np.float32), inputs["500x200_y"].astype(np.float32)


def sampled_error(orc, prob, kernel, X, gamma, rows):
    """max over the sampled rows of |row of the implicit matvec - float64 oracle| on the scale of the row's summands, in fp32 eps"""
    n = X.shape[0] - 1
    rhs = np.random.default_rng(0).uniform(-1, 1, size=n).astype(np.float32)
    q, QA = prob.q()
    got = prob.matvec(rhs, np.zeros(n, np.float32), 1.0)
    X64, q64, rhs64 = X.astype(np.float64), q.astype(np.float64), rhs.astype(np.float64)
    want = np.zeros(n)
    for r in rows:
        want = orc.matvec_rows(kernel, X64, q64, rhs64, want, float(QA), 1.0, 1.0, int(r), int(r) + 1, degree=3, gamma=gamma, coef0=0.0)
    G = X64[rows] @ X64[:n].T
    if kernel == "rbf":
        sq = np.einsum("ij,ij->i", X64, X64)
        K = np.exp(-gamma * np.maximum(sq[rows, None] + sq[None, :n] - 2.0 * G, 0.0))
    else:
        K = np.abs(G)
    absd = np.abs(rhs64)
    scale = K @ absd + (abs(float(QA)) + np.abs(q64[rows])) * absd.sum() + np.abs(q64) @ absd + absd[rows]
    with np.errstate(invalid="ignore", divide="ignore"):
        return float(np.nanmax(np.abs(got[rows] - want[rows]) / scale)) / np.finfo(np.float32).eps


def main():
    n, d = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50_000, 128)
    orc = oracle_lib.oracle()
    print(f"# {n} x {d} (ref500x200: 500 x 200), fp32; f16 planes accepted up to a row error of 2^-22 = {2.0 ** -22:.2e}; error = max over 48 sampled rows, in fp32 eps of the row's summands")
    print(f"{'data':13s} {'kernel':7s} {'f16 row error':>13s} {'rbf R2':>9s} | {'default mode':16s} {'ms/matvec':>9s} {'error':>9s} | {'forced other split':18s} {'ms/matvec':>9s} {'error':>9s}")
    for name, X, y in data_sets(n, d):
        N = X.shape[0]
        rows = np.sort(np.random.default_rng(3).choice(N - 1, size=min(48, N - 1), replace=False))
        for kernel in ("linear", "rbf"):
            gamma = 1.0 / X.shape[1]
            prm = Parameter(kernel_type=kernel, gamma=gamma)
            cells = []
            first = None
            for forced in (None, "other"):
                opts = Options()
                if forced is not None:
                    if first["gram_mode"] not in (1, 2):
                        cells.append(f"{'-':18s} {'-':>9s} {'-':>9s}")
                        continue
                    opts.set("gram_mode", 1 if first["gram_mode"] == 2 else 2)  # (2 = f16x3 WITHOUT the check: what would have happened without it)
                try:
                    with backend.ResidentProblem(prm, X, options=opts) as prob:
                        # the tile kernel's own time per implicit matvec (HIP events inside the library), over 40 products with a FIXED right-hand side: a CG run on such
                        # data converges within a dozen iterations and then iterates on rounding noise (eps = 1e-30 never stops it), which says nothing about the kernel
                        prob.cg_begin(y, 1e-30)
                        rhs = np.random.default_rng(1).uniform(-1, 1, size=X.shape[0] - 1).astype(np.float32)
                        zero = np.zeros(X.shape[0] - 1, np.float32)
                        for _ in range(8):
                            prob.matvec(rhs, zero, 1.0)
                        i0 = prob.info()
                        for _ in range(40):
                            prob.matvec(rhs, zero, 1.0)
                        i1 = prob.info()
                        timed = i1["matvec_timed"] - i0["matvec_timed"]
                        ms = (i1["matvec_kernel_ms_total"] - i0["matvec_kernel_ms_total"]) / max(timed, 1)
                        info = i1
                        err = sampled_error(orc, prob, kernel, X, gamma, rows)
                except Exception as e:  # noqa: BLE001
                    cells.append(f"{type(e).__name__}: {str(e)[:60]}")
                    continue
                if first is None:
                    first = info
                mode = MODES[info["gram_mode"]] + (" (direct rbf)" if info["rbf_direct"] else "")
                cells.append(f"{mode:{16 if forced is None else 18}s} {ms:9.3f} {err:9.2f}")
            r2 = f"{first['rbf_exponent_scale']:9.2f}" if kernel == "rbf" and first else f"{'-':>9s}"
            fe = f"{first['f16_row_rel_error']:13.3e}" if first else f"{'?':>13s}"
            print(f"{name:13s} {kernel:7s} {fe} {r2} | " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
