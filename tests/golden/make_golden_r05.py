#!/usr/bin/env python3
"""Round-5 goldens from the REFERENCE's own OpenMP kernels (oracle/_ref/liblssvm_ref.so = src/plssvm/backends/OpenMP/{svm_kernel,q_kernel}.cpp compiled
in place + the CG recipe of csvm.cpp:71-183, see make_golden.py), ONE OpenMP thread (the only reproducible summation order of its `omp atomic` sums):

  fp32_cg_fixed.npz    fixed-LENGTH fp32 solves, max_iter in {1, 2, 3} with eps = 1e-30, of the 18 systems of make_golden_fp32_cg.py:
                     alpha and rho after exactly k CG iterations (csvm.cpp:125-166) -- where "alpha within 1e-4 rel-inf of OpenMP" (BASELINE.json north_star)
                     is attainable in fp32, before the CG recursion amplifies rounding differences (profiles/r04_ref_fp32_self_reproducibility.log).
  pair_matvec.npz    (item 4b)  one implicit matvec at a size that reaches the headline kernel tile_matvec_f32_pair (256-row workgroups run from 64 row
                     blocks = 8 192 points on): 8 704 x 128 rbf fp32 and 8 704 x 256 linear fp32, seeded blobs; stored: q (all n entries), QA_cost, and
                     the result at 512 sampled rows for add = +1 (the full vectors would be data enough, the sample keeps the fixture small).  The same
                     rows from the float64 run of the same kernels as the yardstick of the reference's own fp32 rounding.

Run in the build container only (needs /root/reference):   make -C oracle ref && python tests/golden/make_golden_r05.py"""

import os
import sys

os.environ["OMP_NUM_THREADS"] = "1"

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import oracle_lib  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402

PARAM_SETS = {"ref": dict(degree=2, gamma=0.001, coef0=1.0, cost=0.1), "def": dict(degree=3, gamma=None, coef0=0.0, cost=1.0)}
KERNELS = ["linear", "polynomial", "rbf"]
PAIR_CASES = {"rbf8704x128": ("rbf", 8704, 128), "linear8704x256": ("linear", 8704, 256)}
PAIR_SEED, PAIR_RHS_SEED, PAIR_ROWS_SEED, PAIR_NROWS = 11, 77, 5, 512


def pair_inputs(name):
    kernel, N, d = PAIR_CASES[name]
    X, _ = make_blobs_pm1(N, d, seed=PAIR_SEED, dtype=np.float32)
    n = N - 1
    rhs = np.random.default_rng(PAIR_RHS_SEED).uniform(1.0, 2.0, size=n).astype(np.float32)  # the reference's own test recipe: [1, 2) (generic_csvm_tests.hpp:439-493)
    rows = np.sort(np.random.default_rng(PAIR_ROWS_SEED).choice(n, size=PAIR_NROWS, replace=False))
    rows[0], rows[-1] = 0, n - 1
    return kernel, X, rhs, rows


def main():
    if not oracle_lib.have_ref():
        raise SystemExit("oracle/_ref/liblssvm_ref.so missing: run `make -C oracle ref` first")
    ref = oracle_lib.ref()
    assert os.environ["OMP_NUM_THREADS"] == "1"  # (set above, before the OpenMP runtime was loaded)
    inp = np.load(os.path.join(HERE, "inputs.npz"))
    sets = {k: (inp[f"{k}_X"], inp[f"{k}_y"]) for k in ("500x200", "blobs263x37")}
    X, y = make_blobs_pm1(2000, 64, seed=5, dtype=np.float64)
    sets["blobs2000x64"] = (X, y.astype(np.float64))
    out = {}
    for name, (X64, y64) in sets.items():
        N, d = X64.shape
        X32, y32 = X64.astype(np.float32), y64.astype(np.float32)
        for kernel in KERNELS:
            for pname, P in PARAM_SETS.items():
                kw = dict(degree=P["degree"], gamma=P["gamma"] if P["gamma"] is not None else 1.0 / d, coef0=P["coef0"])
                for k in (1, 2, 3):
                    key = f"{name}/{kernel}/{pname}/k{k}"
                    a, rho, info = ref.solve(kernel, X32, y32, 1e-30, k, cost=P["cost"], **kw)
                    assert int(info["iterations"]) == k
                    a64, rho64, _ = ref.solve(kernel, X64, y64, 1e-30, k, cost=P["cost"], **kw)
                    out[f"{key}/alpha"], out[f"{key}/rho"] = a, np.asarray(rho, dtype=np.float32)
                    out[f"{key}/alpha64"], out[f"{key}/rho64"] = a64, np.asarray(rho64, dtype=np.float64)
                    print(f"{key:36s} reference fp32 vs its float64 run: alpha {oracle_lib.rel_inf(a, a64):.2e}  rho {abs(float(rho) - float(rho64)):.2e}", flush=True)
    np.savez_compressed(os.path.join(HERE, "fp32_cg_fixed.npz"), **out)
    print("fp32_cg_fixed.npz", os.path.getsize(os.path.join(HERE, "fp32_cg_fixed.npz")), "bytes")

    out = {}
    for name in PAIR_CASES:
        kernel, X, rhs, rows = pair_inputs(name)
        N, d = X.shape
        kw = dict(degree=3, gamma=1.0 / d, coef0=0.0)
        q = ref.q(kernel, X, **kw)
        QA = np.float32(ref.kernel_function(kernel, X[-1], X[-1], **kw)) + np.float32(1.0)
        ret = ref.matvec(kernel, X, q, rhs, np.zeros(N - 1, np.float32), QA, np.float32(1.0), 1.0, **kw)
        X64 = X.astype(np.float64)
        q64 = ref.q(kernel, X64, **kw)
        QA64 = float(ref.kernel_function(kernel, X64[-1], X64[-1], **kw)) + 1.0
        ret64 = ref.matvec(kernel, X64, q64, rhs.astype(np.float64), np.zeros(N - 1), QA64, 1.0, 1.0, **kw)
        out[f"{name}/q"], out[f"{name}/QA_cost"], out[f"{name}/rows"] = q, np.asarray(QA, np.float32), rows
        out[f"{name}/matvec_p1_rows"], out[f"{name}/matvec_p1_rows64"] = ret[rows], ret64[rows]
        out[f"{name}/matvec_p1_absmax"] = np.asarray(np.max(np.abs(ret64)))
        out[f"{name}/X_sha256"] = np.frombuffer(__import__("hashlib").sha256(X.tobytes()).digest(), dtype=np.uint8)
        print(f"{name}: reference fp32 vs its float64 run at the sampled rows: rel-inf {oracle_lib.rel_inf(ret[rows], ret64[rows]):.2e} (= {oracle_lib.rel_inf(ret[rows], ret64[rows]) / np.finfo(np.float32).eps:.1f} eps)", flush=True)
    np.savez_compressed(os.path.join(HERE, "pair_matvec.npz"), **out)
    print("pair_matvec.npz", os.path.getsize(os.path.join(HERE, "pair_matvec.npz")), "bytes")


if __name__ == "__main__":
    main()
