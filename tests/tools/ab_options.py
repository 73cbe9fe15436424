#!/usr/bin/env python3
"""Developer tool (not a test): same-box A/B of library options on one workload shape.

    python tests/tools/ab_options.py --points 300000 --features 128 --kernel rbf --steps 6 --repeat 2 \
        --variant mfma_shape=2 --variant mfma_shape=3 --variant "mfma_shape=3,j_chunk_tiles=32"

Every variant creates a fresh resident problem (the options are snapshotted then), runs `steps` CG iterations after a warm-up and
prints the average tile-kernel time (HIP events on the solver stream) and the wall time per iteration; the variants are interleaved
`repeat` times so that chip-to-chip and thermal drift shows up as spread inside a variant, not as a difference between variants.
Optionally (--check) the first matvec of every variant is compared with the first variant's."""

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from plssvm_amd import _capi, backend  # noqa: E402
from plssvm_amd.datagen import make_blobs_pm1  # noqa: E402
from plssvm_amd.parameter import Parameter  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=300_000)
    ap.add_argument("--features", type=int, default=128)
    ap.add_argument("--kernel", default="rbf")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--variant", action="append", default=[])
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--gamma", type=float, default=None, help="rbf / polynomial gamma (default 1 / features)")
    args = ap.parse_args()
    variants = args.variant or [""]
    dt = np.dtype(args.dtype)
    X, y = make_blobs_pm1(args.points, args.features, seed=42, dtype=dt)
    p = Parameter(kernel_type=args.kernel, gamma=args.gamma)
    defaults = {n: _capi.get_option(n) for n in _capi.OPTION_NAMES + _capi.DEV_OPTION_NAMES}
    ref = None
    v = np.random.default_rng(1).uniform(-1, 1, size=args.points - 1).astype(dt)
    print(f"# {args.points} x {args.features} {args.kernel} {args.dtype}, {args.steps} steps, device {_capi.device_name(0)}", flush=True)
    for rep in range(args.repeat):
        for var in variants:
            for n, val in defaults.items():
                _capi.set_option(n, val)
            env_set = []
            for kv in filter(None, var.split(",")):
                k, val = kv.split("=")
                if k.startswith("ENV:"):  # an environment switch of the library for this variant only (read when the problem is created)
                    os.environ[k[4:]] = val
                    env_set.append(k[4:])
                else:
                    _capi.set_option(k.strip(), int(val))
            with backend.ResidentProblem(p, X) as prob:
                extra = ""
                if args.check and rep == 0:
                    got = prob.matvec(v, np.zeros_like(v), 1.0)
                    if ref is None:
                        ref = got
                    extra = f"  max|diff to first variant| / max|ref| = {np.max(np.abs(got - ref)) / np.max(np.abs(ref)):.2e}"
                prob.cg_begin(y, 1e-30)
                prob.cg_step(args.warmup)
                prob.synchronize()
                i0 = prob.info()
                t0 = time.perf_counter()
                prob.cg_step(args.steps)
                prob.synchronize()
                wall = (time.perf_counter() - t0) / args.steps * 1e3
                i1 = prob.info()
                nt = i1["matvec_timed"] - i0["matvec_timed"]  # (short matvecs are event-bracketed by sampling: average over the timed ones)
                k_ms = (i1["matvec_kernel_ms_total"] - i0["matvec_kernel_ms_total"]) / max(nt, 1)
                for k in env_set:
                    del os.environ[k]
                print(f"rep {rep}  {var or '(defaults)':40s} tile kernel {k_ms:9.4f} ms   iteration {wall:9.4f} ms   sym {i1['symmetric']} gram {i1['gram_mode']}{extra}", flush=True)
    for n, val in defaults.items():
        _capi.set_option(n, val)


if __name__ == "__main__":
    main()
