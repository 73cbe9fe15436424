#!/usr/bin/env python3
"""rbf with a large exponent scale: the tile kernel's time per implicit matvec on grid planes in the 256-row form (round 6, tile_matvec_f32_pair<KT_RBFG>) against the 128-row form
(option mfma_shape = 2: tile_matvec_f32_g6h, round 5), the f16x3 norm expansion (rbf_form = 2: fast but inaccurate at these scales) and the direct kernel (rbf_form = 1).
VERDICT r05 item 7: 50 000 x 128, gamma = 4 from 1.28 ms to <= 1.0 ms.   usage: grid_pair_speed.py > gpurun_out/r06_grid_pair_speed.log"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from plssvm_amd import backend  # noqa: E402
from plssvm_amd._capi import Options  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402


def kernel_ms(prm, X, y, **opts):
    with backend.ResidentProblem(prm, X, options=Options(**opts)) as prob:
        prob.cg_begin(y, 1e-30)
        rhs = np.random.default_rng(1).uniform(-1, 1, size=X.shape[0] - 1).astype(np.float32)
        zero = np.zeros(X.shape[0] - 1, np.float32)
        for _ in range(8):
            prob.matvec(rhs, zero, 1.0)
        i0 = prob.info()
        for _ in range(48):
            prob.matvec(rhs, zero, 1.0)
        i1 = prob.info()
    timed = i1["matvec_timed"] - i0["matvec_timed"]
    return (i1["matvec_kernel_ms_total"] - i0["matvec_kernel_ms_total"]) / max(timed, 1), i1


def main():
    print(f"{'shape':>14s} {'gamma':>6s} {'R2':>7s} | {'256-row grid':>12s} {'128-row grid':>12s} {'f16x3 expansion':>15s} {'direct':>8s}   (ms per implicit matvec, tile kernel)")
    for n, d, gamma in ((50_000, 128, 4.0), (20_000, 128, 4.0), (200_000, 128, 4.0), (50_000, 64, 8.0), (50_000, 100, 2.0)):
        X, y = make_blobs_pm1(n, d, seed=42, dtype=np.float32)
        prm = Parameter(kernel_type="rbf", gamma=gamma)
        t256, info = kernel_ms(prm, X, y)
        assert info["gram_mode"] == 3, info
        t128, _ = kernel_ms(prm, X, y, mfma_shape=2)
        texp, _ = kernel_ms(prm, X, y, rbf_form=2)
        tdir = kernel_ms(prm, X, y, rbf_form=1)[0] if n <= 50_000 else float("nan")
        print(f"{n:>8d}x{d:<5d} {gamma:6.1f} {info['rbf_exponent_scale']:7.0f} | {t256:12.3f} {t128:12.3f} {texp:15.3f} {tdir:8.2f}", flush=True)


if __name__ == "__main__":
    main()
