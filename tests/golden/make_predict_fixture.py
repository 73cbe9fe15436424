#!/usr/bin/env python3
"""Convert the reference's own predict fixtures (tests/data/predict/500x200_test.libsvm, 500x200_{linear,polynomial,rbf}.libsvm.model,
500x200.libsvm.predict -- LIBSVM-trained models and the labels every backend must reproduce exactly,
tests/backends/generic_csvm_tests.hpp:197-247) into one .npz of arrays (data, not source).  Run in the build container only."""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from plssvm_amd.io_libsvm import parse_libsvm_data  # noqa: E402
from plssvm_amd.model import Model  # noqa: E402

REF = "/root/reference/tests/data/predict"
out = {}
X, y = parse_libsvm_data(os.path.join(REF, "500x200_test.libsvm"))
out["test_X"], out["test_y"] = X, np.asarray(y)
out["expected"] = np.array([int(v) for v in open(os.path.join(REF, "500x200.libsvm.predict")).read().split()])
for k in ("linear", "polynomial", "rbf"):
    m = Model.load(os.path.join(REF, f"500x200_{k}.libsvm.model"))
    out[f"{k}_sv"], out[f"{k}_alpha"], out[f"{k}_rho"] = m.support_vectors(), m.alpha, np.asarray(float(m.rho))
    out[f"{k}_labels"] = np.asarray(m.labels())
    out[f"{k}_degree"] = np.asarray(m.params.degree)
    out[f"{k}_gamma"] = np.asarray(np.nan if m.params.gamma is None else m.params.gamma)
    out[f"{k}_coef0"] = np.asarray(m.params.coef0)
    print(k, m.support_vectors().shape, float(m.rho), m.params)
np.savez_compressed(os.path.join(HERE, "predict_500x200.npz"), **out)
print(os.path.getsize(os.path.join(HERE, "predict_500x200.npz")), "bytes")
