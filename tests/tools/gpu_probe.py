#!/usr/bin/env python3
"""Developer probe (not a test): prints the error of the HIP path against the golden vectors / the CPU oracle and a
first timing of the tile kernel.  Run on the GPU box:  python tests/tools/gpu_probe.py [--perf]"""

import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as ol  # noqa: E402
from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402

PARAM_SETS = {"ref": dict(degree=2, gamma=0.001, coef0=1.0, cost=0.1), "def": dict(degree=3, gamma=None, coef0=0.0, cost=1.0)}


def parity():
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden.npz"))
    inp = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
    print("device:", _capi.device_name(0))
    for name in ["5x4", "blobs263x37", "500x200"]:
        X64, y64 = inp[name + "_X"], inp[name + "_y"]
        N, d = X64.shape
        for tag, dt in (("f64", np.float64), ("f32", np.float32)):
            X, y = X64.astype(dt), y64.astype(dt)
            for pn, P in PARAM_SETS.items():
                for k in ["linear", "polynomial", "rbf"]:
                    key = f"{name}/{k}/{tag}/{pn}"
                    prm = Parameter(kernel_type=k, degree=P["degree"], gamma=P["gamma"], coef0=P["coef0"], cost=P["cost"])
                    q = backend.generate_q(prm, X)
                    eq = ol.rel_inf(q, g[key + "/q"])
                    rhs = g[key + "/rhs"]
                    QA = float(g[key + "/QA_cost"])
                    mv = backend.run_device_kernel(prm, q, np.zeros(N - 1, dt), rhs, X, QA, 1.0)
                    e1 = ol.rel_inf(mv, g[key + "/matvec_p1"])
                    mv = backend.run_device_kernel(prm, q, np.zeros(N - 1, dt), rhs, X, QA, -1.0)
                    e2 = ol.rel_inf(mv, g[key + "/matvec_m1"])
                    line = f"{key:36s} q={eq:.1e} mv+={e1:.1e} mv-={e2:.1e}"
                    if key + "/cg_refresh/alpha" in g:
                        for c in ["cg_tight", "cg_refresh", "cg_default"]:
                            a, rho, info = backend.solve_system_of_linear_equations(prm, X, y, float(g[f"{key}/{c}/eps"]), int(g[f"{key}/{c}/max_iter"]))
                            ga = g[f"{key}/{c}/alpha"]
                            # f64 truth for the f32 cases
                            k64 = f"{name}/{k}/f64/{pn}/{c}/alpha"
                            line += f" | {c[3:]}: it {info['iterations']}/{int(g[f'{key}/{c}/iterations'])} a={ol.rel_inf(a, ga):.1e} rho={abs(float(rho) - float(g[f'{key}/{c}/rho'])):.1e}"
                            if tag == "f32":
                                line += f" (vs64 {ol.rel_inf(a, g[k64]):.1e}, ref {ol.rel_inf(ga, g[k64]):.1e})"
                    print(line, flush=True)


def perf(N=50000, d=128, kernel="rbf", dt=np.float32, iters=5):
    X, y = make_blobs_pm1(N, d, seed=42, dtype=dt)
    prm = Parameter(kernel_type=kernel, cost=1.0)
    t0 = time.time()
    prob = backend.ResidentProblem(prm, X)
    print(f"setup {time.time() - t0:.2f}s")
    prob.cg_begin(y, 1e-30)
    t0 = time.time()
    prob.cg_step(iters)
    prob.synchronize()
    t = (time.time() - t0) / iters
    info = prob.info()
    flop = 2.0 * (N - 1) ** 2 * d
    print(f"{kernel} {np.dtype(dt).name} {N}x{d}: {t * 1e3:.2f} ms/iter wall, tile kernel {info['matvec_kernel_ms']:.3f} ms -> {flop / info['matvec_kernel_ms'] / 1e9:.1f} TFLOP/s eff",
          f"delta {info['residuum']:.4e}")
    prob.close()


if __name__ == "__main__":
    if "--split" in sys.argv:
        # opt-in bf16x6 split (gram_mode 1) against the default f32 MFMA path
        for (N, d, kern) in ((50000, 128, "rbf"), (100000, 128, "rbf"), (100000, 128, "linear"), (100000, 64, "polynomial"), (60000, 256, "linear")):
            for gm in (0, 1):
                _capi.set_option("gram_mode", gm)
                print(f"gram_mode={gm}: ", end="")
                perf(N, d, kern, np.float32, iters=6)
        _capi.set_option("gram_mode", 0)
    elif "--wide" in sys.argv:
        # more than 8 k-chunks: v2 kernels with 10..16 chunks against the generic v1 kernel
        for (N, d, kern, dt) in ((60000, 384, "rbf", np.float32), (60000, 512, "linear", np.float32), (40000, 192, "polynomial", np.float64), (40000, 256, "rbf", np.float64)):
            for tk in (0, 1):
                _capi.set_option("tile_kernel", tk)
                print(f"tile_kernel={tk}: ", end="")
                perf(N, d, kern, dt, iters=4)
        _capi.set_option("tile_kernel", 0)
    elif "--small" in sys.argv:
        # column-chunk length for small and mid-size problems (the grid must fill 256 CUs x 2 workgroups)
        for N in (3000, 6400, 12800, 25600):
            for jt in (1, 2, 4, 8, 16):
                _capi.set_option("j_chunk_tiles", jt)
                print(f"N={N} j_chunk_tiles={jt}: ", end="")
                perf(N, 128, "rbf", np.float32, iters=20)
        _capi.set_option("j_chunk_tiles", 0)
    elif "--ablate64" in sys.argv:
        # fp64 v2 kernel (ablation build): 4 = no epilogue, 16 = no LDS-DMA after the prologue, 8 = no barrier
        for kern in ("polynomial", "rbf", "linear"):
            for dbg in (0, 4, 16, 20, 28):
                _capi.set_option("debug_ablate", dbg)
                print(f"debug_ablate={dbg}: ", end="")
                perf(100000, 64, kern, np.float64, iters=4)
        _capi.set_option("debug_ablate", 0)
    elif "--ablate2" in sys.argv:
        for dbg in (0, 1, 4, 5, 16, 20, 28):
            _capi.set_option("debug_ablate", dbg)
            print(f"debug_ablate={dbg}: ", end="")
            perf(100000, 128, "rbf", np.float32, iters=4)
        _capi.set_option("debug_ablate", 0)
    elif "--ablate" in sys.argv:
        for dbg in (0, 1, 5, 9, 13, 4, 8):
            _capi.set_option("debug_ablate", dbg)
            print(f"debug_ablate={dbg}: ", end="")
            perf(50000, 128, "rbf", np.float32, iters=8)
        _capi.set_option("debug_ablate", 0)
    elif "--perf" in sys.argv:
        perf(50000, 128, "rbf", np.float32)
        perf(50000, 128, "linear", np.float32)
        perf(50000, 128, "polynomial", np.float32)
        _capi.set_option("rbf_form", 1)
        perf(50000, 128, "rbf", np.float32)
        _capi.set_option("rbf_form", 0)
        perf(50000, 64, "polynomial", np.float64)
        perf(50000, 64, "rbf", np.float64)
    else:
        parity()
