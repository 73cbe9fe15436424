#!/usr/bin/env python3
"""fp32 CG goldens on the reference's own test parameters (cost = 0.1: well conditioned; /root/reference/tests/backends/generic_csvm_tests.hpp:372-493) and
on the csvm defaults, from the REFERENCE's OpenMP kernels (oracle/_ref/liblssvm_ref.so, see make_golden.py) -- once with ONE OpenMP thread and once
with EIGHT.  The two runs of the same binary differ only in the order in which `omp atomic` adds the partial sums (svm_kernel.cpp:45-51); their
distance is the yardstick for "alpha within 1e-4 rel-inf of OpenMP" in fp32 (BASELINE.json north_star): what the reference does not reproduce of
itself, no other implementation can be asked to reproduce.  Also the float64 solve of the same system.

Run in the build container only (needs /root/reference):   make -C oracle ref && python tests/golden/make_golden_fp32_cg.py
Writes tests/golden/fp32_cg.npz and prints the table committed as profiles/r04_ref_fp32_self_reproducibility.log."""

import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

PARAM_SETS = {"ref": dict(degree=2, gamma=0.001, coef0=1.0, cost=0.1), "def": dict(degree=3, gamma=None, coef0=0.0, cost=1.0)}
KERNELS = ["linear", "polynomial", "rbf"]
EPS = 1e-6


def data_sets():
    from plssvm_amd.datagen import make_blobs_pm1

    inp = np.load(os.path.join(HERE, "inputs.npz"))
    sets = {k: (inp[f"{k}_X"], inp[f"{k}_y"]) for k in ("500x200", "blobs263x37")}
    X, y = make_blobs_pm1(2000, 64, seed=5, dtype=np.float64)
    sets["blobs2000x64"] = (X, y.astype(np.float64))
    return sets


def solve_all(out_file):
    import oracle_lib

    ref = oracle_lib.ref()
    out = {}
    for name, (X64, y64) in data_sets().items():
        N, d = X64.shape
        for kernel in KERNELS:
            for pname, P in PARAM_SETS.items():
                kw = dict(degree=P["degree"], gamma=P["gamma"] if P["gamma"] is not None else 1.0 / d, coef0=P["coef0"])
                key = f"{name}/{kernel}/{pname}"
                a32, r32, i32 = ref.solve(kernel, X64.astype(np.float32), y64.astype(np.float32), EPS, N, cost=P["cost"], **kw)
                out[f"{key}/alpha"] = a32
                out[f"{key}/rho"] = np.asarray(r32, dtype=np.float32)
                out[f"{key}/iterations"] = np.asarray(i32["iterations"], dtype=np.int64)
                out[f"{key}/delta"] = np.asarray(i32["delta"], dtype=np.float64)
                out[f"{key}/delta0"] = np.asarray(i32["delta0"], dtype=np.float64)
                if os.environ.get("OMP_NUM_THREADS") == "1":
                    a64, r64, _ = ref.solve(kernel, X64, y64, 1e-12, N, cost=P["cost"], **kw)
                    out[f"{key}/alpha64"] = a64
                    out[f"{key}/rho64"] = np.asarray(r64, dtype=np.float64)
    np.savez_compressed(out_file, **out)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--worker":
        solve_all(sys.argv[2])
        return
    tmp = {}
    for threads in (1, 8):
        tmp[threads] = f"/tmp/fp32_cg_t{threads}.npz"
        env = dict(os.environ, OMP_NUM_THREADS=str(threads))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", tmp[threads]], check=True, env=env)
    t1, t8 = np.load(tmp[1]), np.load(tmp[8])
    out = {}
    rel = lambda a, b: float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))) / np.max(np.abs(b.astype(np.float64))))  # noqa: E731
    print(f"# fp32 CG of the reference's OpenMP kernels, eps = {EPS:g}: one thread against eight threads of the same binary, and against its float64 solve")
    for k in t1.files:
        out["t1/" + k] = t1[k]
        if not k.endswith("64"):
            out["t8/" + k] = t8[k]
    for k in sorted({k.rsplit("/", 1)[0] for k in t1.files}):
        print(f"{k:32s} iterations {int(t1[k + '/iterations'])} / {int(t8[k + '/iterations'])}   alpha: 1 vs 8 threads {rel(t8[k + '/alpha'], t1[k + '/alpha']):.2e}   1 thread vs float64 {rel(t1[k + '/alpha'], t1[k + '/alpha64']):.2e}"
              f"   rho: 1 vs 8 threads {abs(float(t1[k + '/rho']) - float(t8[k + '/rho'])):.2e}")
    np.savez_compressed(os.path.join(HERE, "fp32_cg.npz"), **out)
    print("fp32_cg.npz", os.path.getsize(os.path.join(HERE, "fp32_cg.npz")), "bytes")


if __name__ == "__main__":
    main()
