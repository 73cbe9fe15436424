#!/usr/bin/env python3
"""Developer probe: predict_values on wide data (rows = points to predict, columns = support vectors; the full-square variant of the panels-inside-a-tile
kernels) -- wall time of the call (upload, set-up, tile kernel, download) over a few repeats.  usage: predict_wide_probe.py [P S d kernel]..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

args = sys.argv[1:] or ["40000", "40000", "640", "rbf", "40000", "40000", "640", "polynomial", "30000", "30000", "2000", "rbf"]
for i in range(0, len(args), 4):
    P, S, d, kernel = int(args[i]), int(args[i + 1]), int(args[i + 2]), args[i + 3]
    Xs, _ = make_blobs_pm1(S, d, seed=1, dtype=np.float32)
    Xp, _ = make_blobs_pm1(P, d, seed=2, dtype=np.float32)
    alpha = np.random.default_rng(3).uniform(-1, 1, S).astype(np.float32)
    p = Parameter(kernel_type=kernel, gamma=1.0 / d, degree=3, coef0=0.5)
    times = []
    out = None
    for rep in range(4):
        t0 = time.perf_counter()
        out = backend.predict_values(p, Xs, alpha, 0.25, None, Xp)
        out = out[0] if isinstance(out, tuple) else out
        times.append(time.perf_counter() - t0)
    print(f"predict {P} points x {S} support vectors x {d} {kernel}: {min(times[1:]) * 1e3:9.2f} ms per call (best of 3 after a warm-up), checksum {float(np.sum(out.astype(np.float64))):.6e}", flush=True)
