#!/usr/bin/env python3
"""numpy model of the rbf kernel on GRID planes (KT_RBFG; plssvm_amd/csrc/lssvm_tile_f32_split.hip.hpp) -- the arithmetic of the tile kernel restated with
float16 planes, exact products and ONE fp32 rounding per 32-feature MFMA, against three yardsticks on the same fp32 data: the float64 kernel matrix, the norm expansion
c_i + c_j + x_i.x_j in fp32 (what the f16x3 kernels evaluate) and the formula-exact sum of (x_i - x_j)^2 in fp32 (the direct vector-ALU kernel).

    python tests/tools/grid_planes_model.py          prints the table quoted in DESIGN.md

tests/test_host_logic.py runs `model()` on one small case (no GPU needed): the h.h chain must come out EXACT for every pair that matters and the grid planes must
land within a small factor of the direct form."""
import math

import numpy as np


def grid_parameters(r2: float):
    """g and sigma as Problem<float>'s constructor chooses them from the exponent scale R2 = max |x'|^2 alone"""
    r2 = max(r2, 1.0)
    g = 2.0 ** math.ceil(math.log2(max(math.sqrt((r2 + 160.0) * 2.0 ** -23), math.sqrt(r2) / 2048.0)))
    sigma = 2.0 ** math.floor(math.log2(60000.0 / (math.sqrt(r2) + g)))
    return g, sigma


def chain(acc, A, B):
    """acc += A B^T the way the matrix cores accumulate: per 32-feature block the 32 products and their sum exact (f16 x f16 in fp32; modelled in float64), one fp32 rounding per block"""
    A, B = A.astype(np.float64), B.astype(np.float64)
    for k in range(0, A.shape[1], 32):
        acc = (acc.astype(np.float64) + A[:, k:k + 32] @ B[:, k:k + 32].T).astype(np.float32)
    return acc


def model(X, gamma):
    """returns a dict: exponent scale, g, sigma, whether the h.h chain was exact on every pair with |t| <= 150, and per form (max relative error of K over the pairs with
    K > 1e-4, max row-sum error on the scale of the row's summands), both in units of the fp32 eps"""
    X64 = X.astype(np.float64)
    xp = ((X64 - X64.mean(0)) * math.sqrt(2 * gamma * math.log2(math.e))).astype(np.float32)  # centred, pre-scaled: what the tile kernels see
    xp64 = xp.astype(np.float64)
    sq = np.sum(xp64 ** 2, 1)
    r2 = float(sq.max())
    t_true = -(0.5 * (sq[:, None] + sq[None, :]) - xp64 @ xp64.T)
    K_true = np.exp2(t_true)
    # norm expansion in fp32
    c = (-0.5 * sq).astype(np.float32)
    K_norm = np.exp2(chain((c[:, None] + c[None, :]).astype(np.float32), xp64, xp64).astype(np.float64))
    # direct form in fp32
    d2 = np.zeros((len(X), len(X)), np.float32)
    for k in range(xp.shape[1]):
        diff = (xp[:, None, k] - xp[None, :, k]).astype(np.float32)
        d2 = (d2 + diff * diff).astype(np.float32)
    K_dir = np.exp2((-0.5 * d2).astype(np.float64))
    # grid planes
    g, sigma = grid_parameters(r2)
    h = np.rint(xp64 / g) * g
    s = xp64 - h
    P0 = (h * sigma).astype(np.float16)
    assert np.array_equal(P0.astype(np.float64), h * sigma), "the grid plane must be exact in f16"
    s1 = (s * sigma).astype(np.float16)
    s2 = ((s * sigma) - s1.astype(np.float64)).astype(np.float16)
    ch = -0.5 * np.sum(h * h, 1)
    e = (-0.5 * sq) - ch
    start = sigma * sigma * (ch[:, None] + ch[None, :])
    acc = start.astype(np.float32)
    assert np.array_equal(acc.astype(np.float64), start), "the start values must be exact in fp32"
    acc = chain(acc, P0, P0)                                                       # phase 0: h x h
    exact = sigma * sigma * (-0.5 * ((h[:, None, :] - h[None, :, :]) ** 2).sum(2))
    relevant = np.abs(t_true) <= 150
    hh_exact = bool(np.array_equal(acc.astype(np.float64)[relevant], exact[relevant]))
    for rows, cols in ((s1, P0), (s2, P0), (P0, s1), (s1, s1), (P0, s2)):          # phases 1-3 (row plane, column plane)
        acc = chain(acc, rows, cols)
    tg = (acc.astype(np.float64) / (sigma * sigma)).astype(np.float32).astype(np.float64)
    E = np.exp2(e).astype(np.float32).astype(np.float64)
    K_grid = np.exp2(tg).astype(np.float32).astype(np.float64) * E[:, None] * E[None, :]
    eps = 2.0 ** -23
    near = K_true > 1e-4

    def stats(K):
        return float(np.max(np.abs(K[near] - K_true[near]) / K_true[near]) / eps), float(np.max(np.abs(K - K_true).sum(1) / K_true.sum(1)) / eps)

    return {"r2": r2, "g": g, "sigma": sigma, "hh_exact": hh_exact, "norm expansion": stats(K_norm), "direct": stats(K_dir), "grid planes": stats(K_grid)}


def clustered(rng, n, d, spread, centres=8, box=1.0):
    C = rng.uniform(-box, box, size=(centres, d))
    return (C[rng.integers(0, centres, n)] + spread * rng.standard_normal((n, d))).astype(np.float32)


def main():
    rng = np.random.default_rng(3)
    for label, X, gamma in (("clusters, gamma 1", clustered(rng, 384, 128, 0.02), 1.0), ("clusters, gamma 10", clustered(rng, 384, 128, 0.02), 10.0),
                            ("tight clusters, gamma 100", clustered(rng, 384, 128, 0.005), 100.0), ("unscaled box of 20, gamma 0.05", clustered(rng, 384, 64, 0.3, box=20.0), 0.05),
                            ("default gamma", clustered(rng, 384, 128, 0.05), 1.0 / 128)):
        r = model(X, gamma)
        print(f"{label}: {X.shape[0]} x {X.shape[1]}, exponent scale {r['r2']:.0f}, g = 2^{int(math.log2(r['g']))}, sigma = 2^{int(math.log2(r['sigma']))}, h.h chain exact where |t| <= 150: {r['hh_exact']}")
        for name in ("norm expansion", "direct", "grid planes"):
            print(f"      {name:16s} max relative error of K over the near pairs {r[name][0]:9.2f} eps   row sums on the scale of their summands {r[name][1]:9.3f} eps")


if __name__ == "__main__":
    main()
