import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter

def run(X, y, p, devices, steps):
    out = []
    with (backend.ResidentProblem(p, X, devices=devices) if devices else backend.ResidentProblem(p, X)) as prob:
        prob.cg_begin(y, 1e-30)
        out.append(prob.info()["residuum"])
        for k in range(steps):
            prob.cg_step(1)
            out.append(prob.info()["residuum"])
        a, rho, info = prob.cg_finish()
    return np.array(out), a

for kernel, dtype, N, d in [("rbf", np.float64, 1300, 40), ("polynomial", np.float64, 2500, 64), ("rbf", np.float64, 2500, 64), ("linear", np.float64, 1300, 40)]:
    X, y = make_blobs_pm1(N, d, seed=33, dtype=dtype)
    p = Parameter(kernel_type=kernel, cost=2.0)
    r1, a1 = run(X, y, p, None, 6)
    r1b, a1b = run(X, y, p, None, 6)
    r2, a2 = run(X, y, p, [0, 0], 6)
    r3, a3 = run(X, y, p, [0, 0, 0], 6)
    print(kernel, N, d)
    print("  single  ", r1)
    print("  single' ", r1b, np.max(np.abs(a1 - a1b)) / np.max(np.abs(a1)))
    print("  2 shards", r2, np.max(np.abs(a2 - a1)) / np.max(np.abs(a1)))
    print("  3 shards", r3, np.max(np.abs(a3 - a1)) / np.max(np.abs(a1)))
