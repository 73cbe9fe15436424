#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE's own OpenMP kernels.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

The expected outputs come from oracle/_ref/liblssvm_ref.so = the reference's src/plssvm/backends/OpenMP/{svm_kernel,
q_kernel}.cpp compiled in place, driven by the CG recipe of src/plssvm/backends/OpenMP/csvm.cpp:71-183 (oracle/ref_shim.cpp).
Inputs: the reference's own data files tests/data/libsvm/5x4.libsvm and 500x200.libsvm (parsed to arrays; data, not
source) plus one seeded synthetic set with ragged sizes.  The fixtures are plain .npz files (inputs + expected outputs).

Cases per (data set, kernel, dtype, parameter set):
  q, QA_cost, one implicit matvec for add = +1 and add = -1 with a seeded right-hand side in [1, 2)
  (the reference's own test recipe, tests/backends/generic_csvm_tests.hpp:439-493), and three CG solves:
    cg_tight   eps = 1e-10 (f64) / 1e-5 (f32), max_iter = N
    cg_refresh eps = 1e-30, max_iter = 60      (forces the iteration-49 residual refresh, csvm.cpp:140-145)
    cg_default eps = 1e-3,  max_iter = N       (csvm.hpp:268-269 defaults)
"""

import os
import sys

# one OpenMP thread: the reference pushes its partial sums with "omp atomic" (svm_kernel.cpp:45-51), so only a
# single-threaded run has a reproducible summation order (the fixtures then regenerate bit-for-bit)
os.environ["OMP_NUM_THREADS"] = "1"

import numpy as np  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.io_libsvm import parse_libsvm_data  # noqa: E402

REF_DATA = "/root/reference/tests/data/libsvm"

# parameter sets: "ref" = the reference's kernel-test parameters (generic_csvm_tests.hpp:372-493), "def" = csvm defaults
PARAM_SETS = {
    "ref": dict(degree=2, gamma=0.001, coef0=1.0, cost=0.1),
    "def": dict(degree=3, gamma=None, coef0=0.0, cost=1.0),  # gamma = 1 / num_features (csvm.hpp:303-307)
}
KERNELS = ["linear", "polynomial", "rbf"]


def independent_parse(filename, num_features):
    """A second, deliberately naive reading of a LIBSVM file (str.split + float), independent of plssvm_amd.io_libsvm: the fixtures
    must not depend on the package's own parser being right (circular for row f1)."""
    rows, labels = [], []
    for line in open(filename):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        tok = line.split()
        labels.append(float(tok[0]))
        row = [0.0] * num_features
        for t in tok[1:]:
            idx, val = t.split(":")
            row[int(idx) - 1] = float(val)
        rows.append(row)
    return np.array(rows, dtype=np.float64), np.array(labels, dtype=np.float64)


def load_inputs():
    sets = {}
    X, y = parse_libsvm_data(os.path.join(REF_DATA, "5x4.libsvm"), dtype=np.float64)
    sets["5x4"] = (X, np.asarray(y, dtype=np.float64))
    X, y = parse_libsvm_data(os.path.join(REF_DATA, "500x200.libsvm"), dtype=np.float64)
    sets["500x200"] = (X, np.asarray(y, dtype=np.float64))
    for name, fname in (("5x4", "5x4.libsvm"), ("500x200", "500x200.libsvm")):
        Xi, yi = independent_parse(os.path.join(REF_DATA, fname), sets[name][0].shape[1])
        assert np.array_equal(Xi, sets[name][0]) and np.array_equal(yi, sets[name][1]), f"{fname}: the two parsers disagree"
    X, y = make_blobs_pm1(263, 37, seed=7, dtype=np.float64)
    sets["blobs263x37"] = (X, y.astype(np.float64))
    return sets


def main():
    if "--verify-inputs" in sys.argv:  # the committed inputs.npz against both parsers, without regenerating anything
        committed = np.load(os.path.join(HERE, "inputs.npz"))
        for k, (X, y) in load_inputs().items():
            assert np.array_equal(committed[f"{k}_X"], X) and np.array_equal(committed[f"{k}_y"], y), k
        print("inputs.npz verified against the package parser AND the independent parser")
        return
    if not oracle_lib.have_ref():
        raise SystemExit("oracle/_ref/liblssvm_ref.so missing: run `make -C oracle ref` first")
    ref = oracle_lib.ref()
    inputs = load_inputs()
    np.savez_compressed(os.path.join(HERE, "inputs.npz"), **{f"{k}_X": v[0] for k, v in inputs.items()},
                        **{f"{k}_y": v[1] for k, v in inputs.items()})

    out = {}
    for name, (X64, y64) in inputs.items():
        N, d = X64.shape
        for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
            X = X64.astype(dt)
            y = y64.astype(dt)
            rng = np.random.default_rng(1234)
            rhs = rng.uniform(1.0, 2.0, size=N - 1).astype(dt)
            for pname, P in PARAM_SETS.items():
                gamma = P["gamma"] if P["gamma"] is not None else 1.0 / d
                kw = dict(degree=P["degree"], gamma=gamma, coef0=P["coef0"])
                for kernel in KERNELS:
                    key = f"{name}/{kernel}/{tag}/{pname}"
                    q = ref.q(kernel, X, **kw)
                    QA_cost = dt(ref.kernel_function(kernel, X[-1], X[-1], **kw)) + dt(1.0) / dt(P["cost"])
                    out[f"{key}/q"] = q
                    out[f"{key}/QA_cost"] = np.asarray(QA_cost, dtype=dt)
                    out[f"{key}/rhs"] = rhs
                    for add, atag in ((1.0, "p1"), (-1.0, "m1")):
                        ret = ref.matvec(kernel, X, q, rhs, np.zeros(N - 1, dtype=dt), QA_cost, dt(1.0) / dt(P["cost"]), add, **kw)
                        out[f"{key}/matvec_{atag}"] = ret
                    if pname == "ref" and name == "500x200":
                        continue  # cost = 0.1 solves add nothing beyond the kernel-level vectors
                    cg_cases = {
                        "cg_tight": dict(eps=1e-10 if dt == np.float64 else 1e-5, max_iter=N),
                        "cg_refresh": dict(eps=1e-30, max_iter=60),
                        "cg_default": dict(eps=1e-3, max_iter=N),
                    }
                    for cname, cc in cg_cases.items():
                        alpha, rho, info, trace = ref.solve(kernel, X, y, cc["eps"], cc["max_iter"], cost=P["cost"], trace=True, **kw)
                        out[f"{key}/{cname}/alpha"] = alpha
                        out[f"{key}/{cname}/rho"] = np.asarray(rho, dtype=dt)
                        out[f"{key}/{cname}/iterations"] = np.asarray(info["iterations"], dtype=np.int64)
                        out[f"{key}/{cname}/delta"] = np.asarray(info["delta"], dtype=np.float64)
                        out[f"{key}/{cname}/delta0"] = np.asarray(info["delta0"], dtype=np.float64)
                        out[f"{key}/{cname}/trace"] = trace
                        out[f"{key}/{cname}/eps"] = np.asarray(cc["eps"], dtype=np.float64)
                        out[f"{key}/{cname}/max_iter"] = np.asarray(cc["max_iter"], dtype=np.int64)
                        print(f"{key:45s} {cname:10s} its={info['iterations']:4d} delta={info['delta']:.6e} delta0={info['delta0']:.6e} rho={float(rho):+.9f}")
    np.savez_compressed(os.path.join(HERE, "golden.npz"), **out)
    for f in ("inputs.npz", "golden.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
