import sys, time, numpy as np
sys.path.insert(0, '.')
from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter
for n in (20000, 50000, 100000, 200000):
    X, y = make_blobs_pm1(n, 128, seed=1, dtype=np.float32)
    for head in (0, 1, 0, 1):
        _capi.set_option("j_chunk_head", head)
        t0 = time.perf_counter()
        with backend.ResidentProblem(Parameter(kernel_type="rbf"), X) as prob:
            t1 = time.perf_counter()
            info = prob.info()
        print(f"{n} points, j_chunk_head {head}: problem created in {1e3 * (t1 - t0):.1f} ms (setup_ms {info['setup_ms']:.1f})", flush=True)
