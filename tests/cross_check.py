"""Randomised cross-check of the tile kernels against the float64 product and the generic kernel -- shared by the driver-run test
(tests/test_gpu_random_cross_check.py: a seeded, bounded slice on every `pytest -m gpu`) and the long-running developer tools
(tests/tools/narrow_stress.py, tests/tools/wide_stress.py).  The reference runs its kernel tests on every backend in every build
(/root/reference/tests/backends/generic_csvm_tests.hpp:372-493); this is the counterpart for the kernel ZOO of this backend: a case draws
shape, feature count, kernel, degree, real type, Gram mode, MFMA-group form, chunk length, band size, item order, record form, shard count and
variant at random, so that instantiations no fixed-shape test names still run (round 3: it found the one wrong instantiation of the round).

A case is a plain dict (reproducible from (family, seed, index) alone); `run_case` returns the error of the chosen kernels and of the generic
kernel against the float64 product, in units of eps of each row's summands, and whether the case passes:
    finite, and  err < max(4 x the generic kernel's own error, 16 eps (fp32) / 256 eps (fp64))."""

from __future__ import annotations

import numpy as np

from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter

OPTION_KEYS = ("gram_mode", "tile_kernel", "j_chunk_tiles", "symmetric", "colslab_band_mb", "rbf_fold", "mfma_shape")


def narrow_case(seed: int, index: int) -> dict:
    """the resident-row-panel kernels: up to 512 features, both real types, every Gram mode and group form, the linear kernel's panel passes"""
    rng = np.random.default_rng([seed, index, 1])
    dtype = ("float32", "float64")[int(rng.integers(2))]
    kernel = ("rbf", "polynomial", "linear")[int(rng.integers(3))]
    # (from 8 193 points on the narrow symmetric fp32 cases run the 256-row workgroups on block pairs: odd and even numbers of row blocks)
    N = int(rng.choice([2, 3, 100, 128, 129, 130, 257, 258, 385, 640, 1000, 1537, 2500, 4100, 8322, 9001, 12290]))
    d = int(rng.choice([1, 3, 16, 31, 64, 65, 100, 128, 129, 192, 200, 256, 257, 300, 384, 385, 448, 512]))
    if N > 8192 and rng.integers(3):
        d = int(rng.choice([16, 64, 65, 100, 128, 200, 256]))  # mostly where the pair kernels apply (<= 128 features; the linear kernel's panel passes beyond)
    opts = dict(gram_mode=int(rng.choice([3, 3, 1, 0, 2])), j_chunk_tiles=int(rng.choice([0, 0, 1, 2, 3, 7])), symmetric=int(rng.choice([1, 1, 0])),
                colslab_band_mb=int(rng.choice([2048, 1])), rbf_fold=int(rng.choice([1, 0])), mfma_shape=int(rng.choice([3, 3, 2])))
    return dict(family="narrow", dtype=dtype, kernel=kernel, N=N, d=d, opts=opts, shards=int(rng.choice([1, 1, 2, 3, 8])), degree=int(rng.choice([0, 1, 2, 3, 4])),
                gamma=float(rng.choice([1.0, 0.3])) / d, coef0=float(rng.choice([0.0, 1.0])), data_seed=300 + index, v_seed=int(rng.integers(1 << 30)))


def pair_case(seed: int, index: int) -> dict:
    """the 256-row workgroups on block pairs (lssvm_tile_f32_pair.hip.hpp): fp32, symmetric, at most 128 features per pass (the linear kernel's panel
    passes beyond), from 64 row blocks on -- odd and even block counts, chunk lengths from one tile, bands, shards, both plane kinds"""
    rng = np.random.default_rng([seed, index, 4])
    kernel = ("rbf", "polynomial", "linear")[int(rng.integers(3))]
    N = int(rng.integers(8194, 13000))
    d = int(rng.choice([1, 17, 64, 65, 100, 128])) if kernel != "linear" or rng.integers(2) else int(rng.choice([129, 200, 256, 300]))
    opts = dict(gram_mode=int(rng.choice([3, 3, 2, 1])), j_chunk_tiles=int(rng.choice([0, 0, 1, 2, 3, 5, 7])), symmetric=1, colslab_band_mb=int(rng.choice([2048, 1])),
                rbf_fold=1, mfma_shape=3)
    return dict(family="pair", dtype="float32", kernel=kernel, N=N, d=d, opts=opts, shards=int(rng.choice([1, 1, 2, 3, 8])), degree=int(rng.choice([2, 3])),
                gamma=float(rng.choice([1.0, 0.3])) / d, coef0=float(rng.choice([0.0, 1.0])), data_seed=500 + index, v_seed=int(rng.integers(1 << 30)))


def wide_case(seed: int, index: int, f64: bool) -> dict:
    """the panels-inside-a-tile kernels (rbf / polynomial on wide data) and, in fp64, the linear kernel's panel passes"""
    rng = np.random.default_rng([seed, index, 3 if f64 else 2])
    kernel = ("rbf", "polynomial", "linear")[int(rng.integers(3 if f64 else 2))]
    N = int(rng.choice([2, 100, 129, 257, 640, 1000, 1537, 2500, 4100]))
    d = int(rng.choice([257, 300, 320, 449, 512, 577, 1025, 2049] if f64 else [385, 449, 512, 513, 577, 640, 700, 1025, 1500, 2049]))
    if not f64 and kernel == "polynomial" and d <= 512 and rng.integers(2):
        d += 256
    opts = dict(gram_mode=int(rng.choice([3, 1])), j_chunk_tiles=int(rng.choice([0, 1, 2, 3, 7])), symmetric=int(rng.choice([1, 1, 0])),
                colslab_band_mb=int(rng.choice([2048, 1])), rbf_fold=int(rng.choice([1, 0])))
    return dict(family="wide_f64" if f64 else "wide_f32", dtype="float64" if f64 else "float32", kernel=kernel, N=N, d=d, opts=opts, shards=int(rng.choice([1, 1, 2, 3, 8])),
                degree=int(rng.choice([1, 2, 3, 4])), gamma=float(rng.choice([1.0, 0.3])) / d, coef0=float(rng.choice([0.0, 1.0])), data_seed=100 + index, v_seed=int(rng.integers(1 << 30)))


def grid_case(seed: int, index: int) -> dict:
    """rbf with a LARGE exponent scale (round 5): gamma d between 60 and 3 500, i.e. exponent scales of ~60 ... 4 000 on [-1, 1]-scaled data -- the automatic choice is the matrix
    cores on grid planes (tile_matvec_f32_g6h up to 128 features, _g6w up to 384) and the direct kernel beyond the limits; ragged shapes and feature counts, both variants,
    chunk lengths, bands, shards.  The right-hand side is made orthogonal to 1 and q (`orthogonal_v`), so that the comparison sees the kernel matrix and not the rank-1 terms."""
    rng = np.random.default_rng([seed, index, 5])
    N = int(rng.choice([2, 3, 100, 129, 257, 640, 1000, 1537, 2500, 4100, 8322, 9001]))
    d = int(rng.choice([1, 3, 16, 31, 64, 65, 100, 128, 129, 192, 200, 256, 257, 300, 320, 384, 385, 500]))
    opts = dict(gram_mode=int(rng.choice([3, 3, 2, 1])), j_chunk_tiles=int(rng.choice([0, 0, 1, 2, 3, 7])), symmetric=int(rng.choice([1, 1, 0])), colslab_band_mb=int(rng.choice([2048, 1])),
                rbf_fold=int(rng.choice([1, 0])), mfma_shape=int(rng.choice([3, 2])))
    return dict(family="grid", dtype="float32", kernel="rbf", N=N, d=d, opts=opts, shards=int(rng.choice([1, 1, 2, 3, 8])), degree=3,
                gamma=float(rng.choice([60.0, 300.0, 1500.0, 3500.0])) / d, coef0=0.0, data_seed=700 + index, v_seed=int(rng.integers(1 << 30)), orthogonal_v=True)


def grid_pair_case(seed: int, index: int) -> dict:
    """rbf with a large exponent scale in the 256-ROW form (round 6: tile_matvec_f32_pair<KT_RBFG>): from 64 row blocks on, at most 128 features, symmetric variant --
    odd and even block counts, chunk lengths from one tile, bands, shards; gamma d from 60 to 3 500 as grid_case."""
    rng = np.random.default_rng([seed, index, 6])
    N = int(rng.integers(8194, 13000))
    d = int(rng.choice([3, 17, 64, 65, 100, 128]))
    opts = dict(gram_mode=int(rng.choice([3, 3, 1])), j_chunk_tiles=int(rng.choice([0, 0, 1, 2, 3, 5, 7])), symmetric=1, colslab_band_mb=int(rng.choice([2048, 1])), rbf_fold=int(rng.choice([1, 0])), mfma_shape=3)
    return dict(family="grid_pair", dtype="float32", kernel="rbf", N=N, d=d, opts=opts, shards=int(rng.choice([1, 1, 2, 3, 8])), degree=3,
                gamma=float(rng.choice([60.0, 300.0, 1500.0, 3500.0])) / d, coef0=0.0, data_seed=900 + index, v_seed=int(rng.integers(1 << 30)), orthogonal_v=True)


def rect_case(seed: int, index: int) -> dict:
    """predict_values on the RECTANGULAR 256-row kernel (round 6: tile_matvec_f32_pair_rect): at least 64 row blocks of points, at most 128 features, rbf (folded records)
    and the polynomial kernel of degree 2 / 3, both plane kinds; ragged point and support-vector counts, chunk lengths from one tile."""
    rng = np.random.default_rng([seed, index, 7])
    kernel = ("rbf", "polynomial")[int(rng.integers(2))]
    return dict(family="rect", kernel=kernel, npts=int(rng.integers(8065, 20000)), nsv=int(rng.choice([1, 2, 127, 128, 129, 1000, 2500, 6001])), d=int(rng.choice([1, 17, 64, 65, 100, 128])),
                degree=int(rng.choice([2, 3])), gamma_d=float(rng.choice([1.0, 0.3])), coef0=float(rng.choice([0.0, 1.0])),
                opts=dict(gram_mode=int(rng.choice([3, 3, 2, 1])), j_chunk_tiles=int(rng.choice([0, 0, 1, 2, 5, 9])), mfma_shape=3), data_seed=1100 + index, a_seed=int(rng.integers(1 << 30)))


def run_rect_case(case: dict) -> dict:
    """predict_values against the float64 sum on every point, on the scale of each point's summands; the 128-row kernels (mfma_shape = 2) as the yardstick"""
    from plssvm_amd._capi import Options

    npts, nsv, d = case["npts"], case["nsv"], case["d"]
    X, _ = make_blobs_pm1(nsv + npts, d, seed=case["data_seed"], dtype=np.float32)
    sv, pts = X[:nsv], X[nsv:]
    alpha = np.random.default_rng(case["a_seed"]).standard_normal(nsv).astype(np.float32)
    gamma = case["gamma_d"] / d
    p = Parameter(kernel_type=case["kernel"], gamma=gamma, degree=case["degree"], coef0=case["coef0"], cost=1.0)
    info = {}
    got, _ = backend.predict_values(p, sv, alpha, 0.5, None, pts, options=Options(**case["opts"]), info_out=info)
    ref, _ = backend.predict_values(p, sv, alpha, 0.5, None, pts, options=Options(**{**case["opts"], "mfma_shape": 2}))
    S, P, a64 = sv.astype(np.float64), pts.astype(np.float64), alpha.astype(np.float64)
    G = P @ S.T
    if case["kernel"] == "rbf":
        K = np.exp(-gamma * np.maximum(np.einsum("ij,ij->i", P, P)[:, None] + np.einsum("ij,ij->i", S, S)[None, :] - 2.0 * G, 0.0))
    else:
        K = (gamma * G + case["coef0"]) ** case["degree"]
    truth = K @ a64 - 0.5
    scale = np.abs(K) @ np.abs(a64) + 0.5
    eps = float(np.finfo(np.float32).eps)
    err, err_ref = float(np.max(np.abs(got - truth) / scale)) / eps, float(np.max(np.abs(ref - truth) / scale)) / eps
    return dict(err=err, err_generic=err_ref, ok=bool(np.all(np.isfinite(got))) and err < max(4.0 * err_ref, 16.0), gram_mode=info["gram_mode"], symmetric=0)


def describe(case: dict) -> str:
    if case["family"] == "rect":
        return f"rect {case['kernel']} points {case['npts']} support vectors {case['nsv']} d {case['d']} degree {case['degree']} coef0 {case['coef0']} gamma*d {case['gamma_d']} {case['opts']}"
    return (f"{case['family']} {case['dtype']} {case['kernel']} N {case['N']} d {case['d']} degree {case['degree']} coef0 {case['coef0']} gamma*d {case['gamma'] * case['d']:.1f} "
            f"shards {case['shards']} {case['opts']}")


def run_case(case: dict) -> dict:
    dtype = np.dtype(case["dtype"])
    N, d, kernel, degree = case["N"], case["d"], case["kernel"], case["degree"]
    X, _ = make_blobs_pm1(N, d, seed=case["data_seed"], dtype=dtype.type)
    p = Parameter(kernel_type=kernel, gamma=case["gamma"], degree=degree, coef0=case["coef0"], cost=1.0)
    v = np.random.default_rng(case["v_seed"]).uniform(-1, 1, N - 1).astype(dtype)
    if case.get("orthogonal_v") and N > 3:
        # orthogonal to 1 and to q = k(x_i, x_last) (float64 from the same data), then rounded: the rank-1 terms of Abar v all but vanish
        Xq = X.astype(np.float64)
        if kernel == "rbf":
            qq = np.exp(-p.gamma * np.sum((Xq[:N - 1] - Xq[N - 1]) ** 2, axis=1))
        else:
            qq = Xq[:N - 1] @ Xq[N - 1]
        basis = np.linalg.qr(np.stack([np.ones(N - 1), qq], axis=1))[0]
        v64o = v.astype(np.float64)
        for _ in range(2):
            v64o = v64o - basis @ (basis.T @ v64o)
        v = v64o.astype(dtype)
    zero = np.zeros(N - 1, dtype)
    defaults = {k: _capi.get_option(k) for k in OPTION_KEYS}
    out, info = {}, {}
    try:
        for label, extra in (("tiles", {}), ("generic", {"tile_kernel": 1})):
            for k, val in defaults.items():
                _capi.set_option(k, val)
            for k, val in {**case["opts"], **extra}.items():
                _capi.set_option(k, val)
            with backend.ResidentProblem(p, X, devices=[0] * case["shards"]) as prob:
                out[label] = prob.matvec(v, zero, 1.0).astype(np.float64)
                info[label] = prob.info()
    finally:
        for k, val in defaults.items():
            _capi.set_option(k, val)
    # float64 truth of Abar v = K v + v / C + (QA_cost S - q.v) 1 - S q
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
    eps = float(np.finfo(dtype).eps)
    err_t = float(np.max(np.abs(out["tiles"] - truth) / scale)) / eps
    err_g = float(np.max(np.abs(out["generic"] - truth) / scale)) / eps
    # (fp64: the data carries sqrt(gamma) / the exponent scale, a power amplifies its rounding; 256 eps = 6e-14)
    ok = bool(np.all(np.isfinite(out["tiles"]))) and err_t < max(4.0 * err_g, 256.0 if dtype == np.float64 else 16.0)
    return dict(err=err_t, err_generic=err_g, ok=ok, gram_mode=info["tiles"]["gram_mode"], symmetric=info["tiles"]["symmetric"])
