#!/usr/bin/env python3
"""Load balance of the sharded path, measured on ONE GPU: every rank's share of the c5 matvec (world = 2, 4, 8) is evaluated in
turn with the exchange switched off (option skip_collective) and its tile-kernel time is reported.
usage: rank_balance.py [N] [d] [world ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128
worlds = [int(v) for v in sys.argv[3:]] or [8]
X, y = make_blobs_pm1(N, d, seed=42, dtype=np.float32)
p = Parameter(kernel_type="rbf")
with backend.ResidentProblem(p, X) as prob:
    prob.cg_begin(y, 1e-30)
    prob.cg_step(3)
    prob.synchronize()
    single = prob.info()["matvec_kernel_ms"]
print(f"{N}x{d} rbf fp32, one GPU, whole problem: tile kernel {single:.2f} ms", flush=True)
_capi.set_option("skip_collective", 1)
for world in worlds:
    times = []
    for rank in range(world):
        with backend.ResidentProblem(p, X, rank=rank, world=world) as prob:
            prob.cg_begin(y, 1e-30)
            prob.cg_step(3)
            prob.synchronize()
            times.append(prob.info()["matvec_kernel_ms"])
    print(f"world {world}: per-rank tile kernel ms " + " ".join(f"{t:.2f}" for t in times) + f"  | max {max(times):.2f}  ideal {single / world:.2f}  "
          f"efficiency {single / world / max(times):.3f}", flush=True)
_capi.set_option("skip_collective", 0)
