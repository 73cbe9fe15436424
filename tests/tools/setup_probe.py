import sys, time
sys.path.insert(0,'.')
import numpy as np
from plssvm_amd import backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.parameter import Parameter
for n in (50_000, 250_000, 330_000, 1_000_000):
    X,y=make_blobs_pm1(n,128,seed=1,dtype=np.float32)
    for rep in range(2):
        t=time.perf_counter()
        with backend.ResidentProblem(Parameter(kernel_type="rbf"), X) as prob:
            prob.synchronize()
            dt=time.perf_counter()-t
            print(n, "create %.1f ms (library setup_ms %.1f)"%(dt*1e3, prob.info()["setup_ms"]), flush=True)
