#!/usr/bin/env python3
"""Generates tests/golden/c1_500x4.npz: BASELINE.json configs[0] -- "generate_data.py 500x4, linear kernel, fp64, OpenMP backend".

Input: the data set of plssvm_amd.datagen (the seeded restatement of the reference's unseeded utility_scripts/generate_data.py),
written as a LIBSVM text file and parsed back by an INDEPENDENT parser (scikit-learn's load_svmlight_file), i.e. exactly the
values a plssvm-train run would see.  Expected outputs: alpha, rho, iteration count and final residuum of the reference's own
OpenMP kernels (oracle/_ref/liblssvm_ref.so, built from /root/reference by oracle/Makefile; single threaded => bit reproducible)
at the default eps = 1e-3 and at eps = 1e-10.  Run here (needs oracle/_ref); the npz is committed, this script with it."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["OMP_NUM_THREADS"] = "1"

import oracle_lib  # noqa: E402
from plssvm_amd.datagen import generate_libsvm_file  # noqa: E402
from sklearn.datasets import load_svmlight_file  # noqa: E402


def main():
    if not oracle_lib.have_ref():
        raise SystemExit("oracle/_ref/liblssvm_ref.so missing: run `make -C oracle ref` first")
    with tempfile.TemporaryDirectory() as tmp:
        f = os.path.join(tmp, "500x4.libsvm")
        generate_libsvm_file(f, 500, 4, seed=1)
        Xs, y = load_svmlight_file(f, n_features=4, dtype=np.float64, zero_based=False)
        text = open(f).read()
    X = np.ascontiguousarray(Xs.toarray())
    out = {"X": X, "y": y.astype(np.float64), "libsvm_text": np.frombuffer(text.encode(), dtype=np.uint8)}
    ref = oracle_lib.ref()
    for tag, eps in (("default", 1e-3), ("tight", 1e-10)):
        alpha, rho, info, trace = ref.solve("linear", X, y.astype(np.float64), eps, 500, cost=1.0, trace=True)
        out[f"{tag}/eps"] = np.asarray(eps)
        out[f"{tag}/alpha"] = alpha
        out[f"{tag}/rho"] = np.asarray(rho)
        out[f"{tag}/iterations"] = np.asarray(info["iterations"], dtype=np.int64)
        out[f"{tag}/delta"] = np.asarray(info["delta"])
        out[f"{tag}/trace"] = trace
        print(tag, "its", info["iterations"], "delta", info["delta"], "rho", float(rho))
    np.savez_compressed(os.path.join(HERE, "c1_500x4.npz"), **out)
    print(os.path.getsize(os.path.join(HERE, "c1_500x4.npz")), "bytes")


if __name__ == "__main__":
    main()
