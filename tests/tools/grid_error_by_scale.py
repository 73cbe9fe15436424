#!/usr/bin/env python3
"""rbf on grid planes: the error of a row of the implicit matvec over the exponent scale R2, by feature count -- grid planes (rbf_form 3) beside the formula-exact
kernel (rbf_form 1) on the same data, both against the float64 product on the scale of the row's summands (cross_check.run_case's measure).
usage: grid_error_by_scale.py [seeds]"""
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np  # noqa: E402

import cross_check  # noqa: E402
from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
print(f"{'d':>4} {'gamma*d':>8} {'R2':>8} | grid planes: worst / median eps over {seeds} data sets | direct kernel: worst eps | ratio of the worst")
for d in (3, 17, 64, 128):
    for gd in (100.0, 300.0, 700.0, 1000.0, 1500.0, 2000.0, 2500.0, 3000.0, 3500.0):
        eg, ed, r2s = [], [], []
        for k in range(seeds):
            N = 9001 + 37 * k
            case = dict(family="grid_pair", dtype="float32", kernel="rbf", N=N, d=d, opts=dict(gram_mode=3, symmetric=1, mfma_shape=3), shards=1, degree=3, gamma=gd / d, coef0=0.0,
                        data_seed=4000 + 10 * k + d, v_seed=77 + k, orthogonal_v=True)
            X, _ = make_blobs_pm1(N, d, seed=case["data_seed"], dtype=np.float32)
            _capi.set_option("rbf_form", 0)
            with backend.ResidentProblem(Parameter(kernel_type="rbf", gamma=case["gamma"]), X) as prob:
                info = prob.info()
            r2s.append(info["rbf_exponent_scale"])
            try:
                _capi.set_option("rbf_form", 3)
                rg = cross_check.run_case(case)
                _capi.set_option("rbf_form", 1)
                rd = cross_check.run_case(case)
            except Exception as e:  # noqa: BLE001  (beyond the planes' range rbf_form 3 reports an error)
                eg.append(float("nan"))
                ed.append(float("nan"))
                continue
            finally:
                _capi.set_option("rbf_form", 0)
            eg.append(rg["err"] if rg["gram_mode"] == 3 else float("nan"))
            ed.append(rd["err"])
        print(f"{d:4d} {gd:8.0f} {np.mean(r2s):8.0f} | {np.nanmax(eg):8.2f} / {np.nanmedian(eg):8.2f} | {np.nanmax(ed):8.2f} | {np.nanmax(eg) / max(np.nanmax(ed), 1e-9):6.2f}", flush=True)
