#!/usr/bin/env python3
"""One case of tests/tools/grid_stress.py again, in the 256-row form, the 128-row form (mfma_shape 2), one shard and one band, with the exponent scale the library measured.
usage: repro_grid_case.py family seed index"""
import os
import sys

sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import copy  # noqa: E402

import numpy as np  # noqa: E402

import cross_check  # noqa: E402
from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

family, seed, index = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
case = {"grid": cross_check.grid_case, "grid_pair": cross_check.grid_pair_case}[family](seed, index)
print(case)
X, _ = make_blobs_pm1(case["N"], case["d"], seed=case["data_seed"], dtype=np.float32)
with backend.ResidentProblem(Parameter(kernel_type="rbf", gamma=case["gamma"]), X) as prob:
    print("library info:", {k: prob.info()[k] for k in ("gram_mode", "rbf_direct", "symmetric")})
for label, change in (("as drawn", {}), ("128-row form", {"mfma_shape": 2}), ("one shard", {"_shards": 1}), ("one band", {"colslab_band_mb": 2048}), ("one shard, one band, 128-row", {"_shards": 1, "colslab_band_mb": 2048, "mfma_shape": 2}),
                      ("direct kernel", {"rbf_form": 1})):
    c = copy.deepcopy(case)
    for k, v in change.items():
        if k == "_shards":
            c["shards"] = v
        else:
            c["opts"][k] = v
    print(f"{label:32s}", cross_check.run_case(c), flush=True)
