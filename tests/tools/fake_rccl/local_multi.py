#!/usr/bin/env python3
"""ONE process driving several shards (lssvm_mi355_problem_create_multi) over RCCL group calls -- Exchange::local_rccl in
plssvm_amd/csrc/lssvm_exchange.hip -- with all shards on ONE device, through the tests' stand-in for RCCL (the real one refuses repeated
devices).  Started as a fresh child process by tests/test_gpu_fake_rccl.py: the stand-in is loaded FIRST, so the product library's own
dlopen("librccl.so.1") resolves to it by SONAME.  The same problem then runs over the product's peer kernels (exchange = 2: the same fixed
rank-order sum -> the same bits) and on a single shard; everything is written to --out as JSON."""

import argparse
import ctypes
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=4)
    ap.add_argument("--symmetric", type=int, default=1)
    ap.add_argument("--kernel", default="rbf")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--points", type=int, default=6000)
    ap.add_argument("--features", type=int, default=128)
    ap.add_argument("--out", required=True)
    args = ap.parse_args()
    sys.path.insert(0, HERE)
    import preload

    stand_in = preload.load()  # before the product library

    import numpy as np

    from plssvm_amd import _capi, backend
    from plssvm_amd.datagen import make_blobs_pm1
    from plssvm_amd.parameter import Parameter

    _capi.set_option("symmetric", args.symmetric)
    if not args.symmetric:
        _capi.set_option("j_chunk_tiles", 2)  # equal chunking for every shard count: the full-square rows associate identically
    dt = np.dtype(args.dtype)
    X, y = make_blobs_pm1(args.points, args.features, seed=5, dtype=dt)
    p = Parameter(kernel_type=args.kernel)
    n = args.points - 1
    v = np.random.default_rng(9).uniform(-1, 1, size=n).astype(dt)
    zero = np.zeros(n, dt)

    def run(devices, exchange):
        _capi.set_option("exchange", exchange)
        with backend.ResidentProblem(p, X, devices=devices) as prob:
            mv = prob.matvec(v, zero, 1.0)
            prob.cg_begin(y, 1e-30)
            prob.cg_step(args.steps)
            alpha, rho, info = prob.cg_finish()
        return mv, alpha, float(rho), info

    devices = [0] * args.shards
    mv1, a1, rho1, i1 = run(devices, 1)  # RCCL (the stand-in)
    mv2, a2, rho2, i2 = run(devices, 2)  # the product's peer kernels
    mv0, a0, rho0, i0 = run([0], 0)      # one shard
    scale = float(np.max(np.abs(mv0)))
    out = {"rccl_library": backend.comm_library_path(), "stand_in_loaded": hasattr(ctypes.CDLL(backend.comm_library_path()), "fake_rccl_marker") and stand_in is not None,
           "exchange": int(i1["exchange"]), "rccl_nranks": int(i1["rccl_nranks"]), "rccl_rank": int(i1["rccl_rank"]), "rccl_device": int(i1["rccl_device"]),
           "local_devices": int(i1["local_devices"]), "devices_used": int(i1["devices_used"]), "symmetric": int(i1["symmetric"]), "iterations": int(i1["iterations"]),
           "peer_exchange": int(i2["exchange"]), "peer_rccl_nranks": int(i2["rccl_nranks"]),
           "matvec_equal_bits_vs_peer": bool(np.array_equal(mv1, mv2)), "alpha_equal_bits_vs_peer": bool(np.array_equal(a1, a2) and rho1 == rho2),
           "matvec_equal_bits_vs_single": bool(np.array_equal(mv1, mv0)), "alpha_equal_bits_vs_single": bool(np.array_equal(a1, a0) and rho1 == rho0),
           "matvec_err_vs_single": float(np.max(np.abs(mv1 - mv0)) / scale), "finite": bool(np.all(np.isfinite(a1)))}
    with open(args.out, "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
