import numpy as np
from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter
def cg_np(A, b, its):
    x = np.ones_like(b); r = b - A @ x; d = r.copy(); delta = r @ r
    for _ in range(its):
        Ad = A @ d; alpha = delta / (d @ Ad); x = x + alpha * d; r = r - alpha * Ad
        dn = r @ r; beta = dn / delta; delta = dn; d = r + beta * d
    return x
for kernel, N, d in (("polynomial", 1500, 700), ("rbf", 2700, 257), ("linear", 900, 2049)):
    X, y = make_blobs_pm1(N, d, seed=13, dtype=np.float64)
    p = Parameter(kernel_type=kernel, degree=3, gamma=1.0 / d, coef0=0.5, cost=2.0)
    n = N - 1
    G = X @ X.T
    if kernel == "linear": Ka = G
    elif kernel == "polynomial": Ka = (G / d + 0.5) ** 3
    else:
        sq = np.einsum("ij,ij->i", X, X); Ka = np.exp(-np.maximum(sq[:, None] + sq[None, :] - 2.0 * G, 0.0) / d)
    K, q, QA = Ka[:n, :n], Ka[:n, n], Ka[n, n] + 0.5
    A = K + 0.5 * np.eye(n) + QA - q[:, None] - q[None, :]
    b = y[:n] - y[n]
    ref = cg_np(A, b, 4)
    out = {}
    for label, opts in (("panels", {}), ("generic", {"tile_kernel": 1}), ("panels, full square", {"symmetric": 0})):
        _capi.set_option("tile_kernel", 0); _capi.set_option("symmetric", 1)
        for k, v in opts.items(): _capi.set_option(k, v)
        with backend.ResidentProblem(p, X) as prob:
            prob.cg_begin(y, 1e-30); prob.cg_step(4); a = prob.cg_finish()[0]
        print(f"{kernel} {N}x{d} {label:20s}: rel-inf distance of alpha[:n] from a numpy float64 CG: {np.max(np.abs(a[:n] - ref)) / np.max(np.abs(ref)):.3e}")
    _capi.set_option("tile_kernel", 0); _capi.set_option("symmetric", 1)
