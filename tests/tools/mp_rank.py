#!/usr/bin/env python3
"""One rank of a one-process-per-GPU run THROUGH THE LIBRARY (tests/test_gpu_multi_device.py starts two of these as fresh child
processes; it also runs under torchrun).  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment.  --exchange 1: the RCCL
unique id is handed over through a gloo process group (CPU), so the only RCCL user in the process is libplssvm_amd.so itself;
--exchange 2: no RCCL at all, the ranks map each other's partial vectors with HIP IPC (works with several ranks on ONE device).  Rank r runs
ResidentProblem(rank=r, world=W): one implicit matvec and a few CG iterations; rank 0 also runs the single-GPU problem and writes
the distances.  Every rank writes a hash of its alpha: all ranks must hold the same bits.

--rccl-stand-in PATH: load the tests' single-device stand-in for RCCL (tests/tools/fake_rccl/librccl.so.1) into THIS process before anything
else, so that the product library's own dlopen("librccl.so.1") resolves to it by SONAME -- the product's RCCL exchange then runs with several ranks
on ONE device (the real RCCL refuses that).  With it and --exchange 1 every rank ALSO repeats the run over HIP IPC + the peer kernel (the same
fixed rank-order sum) and reports whether the two exchanges agree bit for bit."""

import argparse
import ctypes
import hashlib
import json
import os

os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")  # single node: RCCL's bootstrap need not scan the interfaces (it took 100-600 s on some boxes)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--symmetric", type=int, default=1)
    ap.add_argument("--exchange", type=int, default=1, choices=[1, 2])
    ap.add_argument("--kernel", default="rbf")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--points", type=int, default=6000)
    ap.add_argument("--features", type=int, default=128)
    ap.add_argument("--out", required=True)
    ap.add_argument("--rccl-stand-in", default=None)
    ap.add_argument("--rebalance-after", type=int, default=0, help="exchange 1: after this many CG steps the ranks call problem_rebalance (measured shares, gathered over the RCCL communicator) and report what it did")
    args = ap.parse_args()
    stand_in = None
    if args.rccl_stand_in:  # BEFORE torch and the product library
        sys.path.insert(0, os.path.dirname(os.path.abspath(args.rccl_stand_in)))
        import preload

        stand_in = preload.load(args.rccl_stand_in)
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))

    import numpy as np
    import torch.distributed as dist

    from plssvm_amd import _capi, backend
    from plssvm_amd.datagen import make_blobs_pm1
    from plssvm_amd.parameter import Parameter
    from plssvm_amd.sharding import connect_peers, exchange_unique_id

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if args.exchange == 1:
            uid = exchange_unique_id(dist, backend.comm_get_unique_id, device=None)
            backend.comm_init(local, rank, world, uid)
        _capi.set_option("exchange", args.exchange)
        _capi.set_option("symmetric", args.symmetric)
        if not args.symmetric:
            _capi.set_option("j_chunk_tiles", 2)  # equal chunking for every world size: the full-square rows associate identically
        dt = np.dtype(args.dtype)
        X, y = make_blobs_pm1(args.points, args.features, seed=5, dtype=dt)
        p = Parameter(kernel_type=args.kernel)
        n = args.points - 1
        v = np.random.default_rng(9).uniform(-1, 1, size=n).astype(dt)
        zero = np.zeros(n, dt)
        with backend.ResidentProblem(p, X, device=local, rank=rank, world=world) as prob:
            if args.exchange == 2:
                connect_peers(dist, prob)
            got = prob.matvec(v, zero, 1.0)
            prob.cg_begin(y, 1e-30)
            rebalanced = after = None
            if args.rebalance_after > 0 and args.exchange == 1:
                prob.cg_step(args.rebalance_after)
                rebalanced = prob.rebalance()                      # by measured pace: every rank gathers every rank's time and computes the same weights
                forced = prob.rebalance([1.0 + 0.5 * r for r in range(world)])  # and explicit ones, the same list on every rank
                after = prob.matvec(v, zero, 1.0)                  # (the scratch vector's product: the CG state is untouched)
                prob.cg_step(args.steps - args.rebalance_after)
            else:
                prob.cg_step(args.steps)
            alpha, rho, info = prob.cg_finish()
            dist.barrier()  # (IPC: a rank's vector stays mapped by its peers until they are done)
        out = {"rank": rank, "alpha_sha": hashlib.sha256(alpha.tobytes()).hexdigest(), "rho": float(rho), "devices_used": int(info["devices_used"]),
               "exchange": int(info["exchange"]), "symmetric": int(info["symmetric"]), "rccl_nranks": int(info["rccl_nranks"]), "rccl_rank": int(info["rccl_rank"]),
               "rccl_device": int(info["rccl_device"]), "iterations": int(info["iterations"])}
        if after is not None:
            out["rebalanced_by_measurement"] = bool(rebalanced)
            out["rebalanced_by_weights"] = bool(forced)
            out["matvec_after_rebalance_sha"] = hashlib.sha256(after.tobytes()).hexdigest()
            out["matvec_after_rebalance_err"] = float(np.max(np.abs(after.astype(np.float64) - got.astype(np.float64))) / np.max(np.abs(got)))
        if args.exchange == 1:
            out["rccl_library"] = backend.comm_library_path()
            out["stand_in_loaded"] = bool(stand_in is not None and hasattr(ctypes.CDLL(out["rccl_library"]), "fake_rccl_marker"))
        if args.exchange == 1 and stand_in is not None and after is None:
            # the same problem over HIP IPC + the peer kernel: the stand-in sums in rank order like k_peer_sum, so the two exchanges must agree bit for bit
            _capi.set_option("exchange", 2)
            with backend.ResidentProblem(p, X, device=local, rank=rank, world=world) as prob2:
                connect_peers(dist, prob2)
                got2 = prob2.matvec(v, zero, 1.0)
                prob2.cg_begin(y, 1e-30)
                prob2.cg_step(args.steps)
                alpha2, rho2, info2 = prob2.cg_finish()
                dist.barrier()
            _capi.set_option("exchange", 1)
            out["peer_exchange_matvec_equal_bits"] = bool(np.array_equal(got, got2))
            out["peer_exchange_alpha_equal_bits"] = bool(np.array_equal(alpha, alpha2) and float(rho) == float(rho2))
            out["peer_exchange"] = int(info2["exchange"])
        if rank == 0:
            with backend.ResidentProblem(p, X, device=local) as single:
                want = single.matvec(v, zero, 1.0)
                single.cg_begin(y, 1e-30)
                single.cg_step(args.steps)
                a1, rho1, _ = single.cg_finish()
            with backend.ResidentProblem(p, X.astype(np.float64), device=local) as truth:  # the same iterations in float64: the yardstick
                truth.cg_begin(y.astype(np.float64), 1e-30)
                truth.cg_step(args.steps)
                a64, _, _ = truth.cg_finish()
            out["matvec_err"] = float(np.max(np.abs(got - want)) / np.max(np.abs(want)))
            out["matvec_equal_bits"] = bool(np.array_equal(got, want))
            out["alpha_equal_bits"] = bool(np.array_equal(alpha, a1))
            out["alpha_err64"] = float(np.max(np.abs(alpha - a64)) / np.max(np.abs(a64)))
            out["single_err64"] = float(np.max(np.abs(a1 - a64)) / np.max(np.abs(a64)))
        dist.barrier()
        if args.exchange == 1:
            backend.comm_destroy()
        with open(args.out, "w") as f:
            json.dump(out, f)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
