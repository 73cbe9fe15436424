#!/usr/bin/env python3
"""Where the time of a resident predictor goes: creation, first and later batches, against the one-shot call (LSSVM_MI355_DEBUG=1 prints the laps inside a call).
usage: predictor_timing.py [num_sv [num_points [points of a solve that runs first in this process, 0 = none]]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402

from plssvm_amd import backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

nsv = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
X, _ = make_blobs_pm1(max(nsv, npts), 128, seed=1, dtype=np.float32)
sv, pts = X[:nsv], X[:npts]
alpha = np.random.default_rng(0).uniform(-1, 1, nsv).astype(np.float32)
p = Parameter(kernel_type="rbf", gamma=1.0 / 128)
big = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if big > 0:  # a process with a history of large allocations, like bench.py's when its end-to-end block runs
    Xb, _ = make_blobs_pm1(big, 128, seed=2, dtype=np.float32)
    with backend.ResidentProblem(p, Xb) as prob:
        prob.q()
        prob.matvec(np.ones(big - 1, np.float32), np.zeros(big - 1, np.float32), 1.0)
    del Xb
    print(f"a solve's set-up and one matvec at {big} points came first", flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    pr = backend.Predictor(p, sv, alpha, 0.1)
    t1 = time.perf_counter()
    info = {}
    v2 = pr.predict(pts, info)
    t2 = time.perf_counter()
    v3 = pr.predict(pts, info)
    t3 = time.perf_counter()
    v4 = pr.predict(pts[:100], info)
    t4 = time.perf_counter()
    pr.close()
    print(f"predictor create {1e3 * (t1 - t0):.2f} ms, first batch {1e3 * (t2 - t1):.2f} ms, second {1e3 * (t3 - t2):.2f} ms, 100 points {1e3 * (t4 - t3):.2f} ms, resident {info['resident']}", flush=True)
try:  # the batch and the values in HBM (LSSVM_MEM_DEVICE): torch holds the tensors
    import torch

    Pd = torch.from_numpy(pts).cuda()
    Od = torch.zeros(npts, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    with backend.Predictor(p, sv, alpha, 0.1) as pr:
        for rep in range(4):
            info = {}
            t0 = time.perf_counter()
            pr.predict_device(Pd.data_ptr(), npts, Od.data_ptr(), info)
            t1 = time.perf_counter()
            print(f"batch and values in HBM: call {1e3 * (t1 - t0):.2f} ms (library {info['total_ms']:.2f} ms, kernel {info['kernel_ms']:.2f} ms), same values {np.array_equal(Od.cpu().numpy(), v2)}",
                  flush=True)
except ImportError:
    pass
for rep in range(3):
    t0 = time.perf_counter()
    v, w = backend.predict_values(p, sv, alpha, 0.1, None, pts)
    t1 = time.perf_counter()
    print(f"one-shot call {1e3 * (t1 - t0):.2f} ms, same values {np.array_equal(v, v2)}", flush=True)
