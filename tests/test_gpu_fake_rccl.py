"""GPU (-m gpu): the product's RCCL exchange with MORE THAN ONE RANK -- on one device, through a stand-in for RCCL.

`Solver<T>::exchange` (plssvm_amd/csrc/lssvm_exchange.hip) combines the row-block shards' partial K*v once per implicit matvec with
ncclAllReduce (symmetric variant) or an in-place ncclAllGather (full square); it replaces the reference's host-staged
gpu_csvm::device_reduction (include/plssvm/backends/gpu_csvm.hpp:449-475, tested there on whatever devices exist by
tests/backends/generic_csvm_tests.hpp:495-540).  The real RCCL refuses two ranks on one device and this pool's boxes have one MI355X, so until
round 5 that code had only ever run with a world of one.  tests/tools/fake_rccl/librccl.so.1 is a TEST-ONLY library with RCCL's SONAME and the
entry points the product binds, RCCL's semantics (stream ordered, asynchronous to the host, in place, group calls), and no objection to several
ranks per device; a CHILD process loads it first, so the product's own dlopen("librccl.so.1") finds it.  The product library is the shipped one,
unchanged, and never looks for the stand-in (tests/test_capi_symbols.py).

What is checked, for worlds of 2, 4 and 8, symmetric and full square, fp32 and fp64, short runs and runs across the iteration-49 residual refresh:
  * one process per rank (lssvm_mi355_comm_init -> Exchange::process_rccl) and one process driving all shards (lssvm_mi355_problem_create_multi
    with exchange = 1 -> Exchange::local_rccl, group calls);
  * ncclCommCount as reported through lssvm_cg_info.rccl_nranks equals the world;
  * every rank ends with the same bits;
  * the result equals the product's OWN peer-kernel exchange bit for bit (the stand-in sums in rank order, like k_peer_sum);
  * the full-square variant equals the single-device run bit for bit (row-owned sums);
  * bench.py --gpus 4 --rank-devices 0,0,0,0 with the default exchange prints a line whose config.rccl_nranks is 4.
"""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import HERE, ROOT

pytestmark = pytest.mark.gpu

STAND_IN = os.path.join(HERE, "tools", "fake_rccl", "librccl.so.1")


def _need_stand_in():
    if not os.path.isfile(STAND_IN):
        pytest.fail(f"{STAND_IN} is not built (python -c 'import __graft_entry__ as g; g.build()')")


def _run_ranks(tmp_path, world, extra, env_extra=None):
    """`world` fresh child processes of tests/tools/mp_rank.py, all on device 0, the stand-in loaded first in each."""
    port = 31000 + (os.getpid() * 7 + len(os.listdir(tmp_path)) * 13 + world) % 2000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   FAKE_RCCL_TIMEOUT_S="60", **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "tools", "mp_rank.py"), "--out", str(tmp_path / f"rank{r}.json"), "--exchange", "1",
                                       "--rccl-stand-in", STAND_IN, *extra], env=env, cwd=ROOT))
    codes = []
    try:
        for pr in procs:
            codes.append(pr.wait(timeout=420))
    finally:
        for pr in procs:  # (exactly the processes started above)
            if pr.poll() is None:
                pr.kill()
    assert codes == [0] * world
    return [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]


@pytest.mark.parametrize("world, sym, kernel, dtype, steps", [(2, 1, "rbf", "float32", 8), (2, 0, "rbf", "float32", 8), (4, 1, "linear", "float32", 55), (4, 0, "polynomial", "float64", 3),
                                                              (8, 1, "rbf", "float32", 55), (8, 0, "linear", "float32", 55), (3, 1, "polynomial", "float64", 3)])
def test_one_process_per_rank_over_the_rccl_exchange(tmp_path, world, sym, kernel, dtype, steps):
    """lssvm_mi355_comm_init on every rank, then ResidentProblem(rank, world) with exchange = 1: Exchange::process_rccl -- ncclAllReduce of
    num_tiles * 128 reals in place (symmetric), ncclAllGather of the rank's slice in place at Kv + rank * slice (full square) -- on the solver's stream,
    55 steps cross the residual refresh of iteration 49 (csvm.cpp:131-140)."""
    _need_stand_in()
    res = _run_ranks(tmp_path, world, ["--symmetric", str(sym), "--kernel", kernel, "--dtype", dtype, "--points", "5000", "--features", "96", "--steps", str(steps)])
    assert all(r["stand_in_loaded"] and os.path.samefile(r["rccl_library"], STAND_IN) for r in res)
    assert all(r["exchange"] == 1 and r["devices_used"] == world and r["symmetric"] == sym and r["iterations"] == steps for r in res)
    assert [r["rccl_nranks"] for r in res] == [world] * world and [r["rccl_rank"] for r in res] == list(range(world)) and all(r["rccl_device"] == 0 for r in res)
    assert len({r["alpha_sha"] for r in res}) == 1 and len({r["rho"] for r in res}) == 1  # every rank holds the same bits
    # ... and they are the bits of the product's own peer-kernel exchange (same partition, same rank-order sum)
    assert all(r["peer_exchange"] == 2 and r["peer_exchange_matvec_equal_bits"] and r["peer_exchange_alpha_equal_bits"] for r in res)
    eps = np.finfo(np.dtype(dtype)).eps
    if sym:
        assert res[0]["matvec_err"] < 64 * eps
        if steps <= 8:
            assert res[0]["alpha_err64"] < 2 * res[0]["single_err64"] + (1e-4 if dtype == "float32" else 1e-8)
    else:
        assert res[0]["matvec_equal_bits"] and res[0]["alpha_equal_bits"]  # row-owned sums: the single-device bits


def test_rccl_exchange_from_the_calling_thread(tmp_path):
    """The stand-in's other mode (FAKE_RCCL_SYNC=1: the waits run in the calling thread after a stream synchronisation instead of in host functions
    of the stream): the product's results must not depend on WHEN the collective's host side runs."""
    _need_stand_in()
    res = _run_ranks(tmp_path, 4, ["--symmetric", "1", "--kernel", "rbf", "--dtype", "float32", "--points", "5000", "--features", "96", "--steps", "8"], {"FAKE_RCCL_SYNC": "1"})
    assert [r["rccl_nranks"] for r in res] == [4] * 4 and len({r["alpha_sha"] for r in res}) == 1
    assert all(r["peer_exchange_matvec_equal_bits"] and r["peer_exchange_alpha_equal_bits"] for r in res)


@pytest.mark.parametrize("shards, sym, kernel, dtype, steps", [(2, 1, "rbf", "float32", 8), (4, 1, "rbf", "float32", 55), (4, 0, "linear", "float32", 55), (8, 1, "polynomial", "float64", 3),
                                                               (8, 0, "rbf", "float32", 8), (3, 0, "polynomial", "float64", 3)])
def test_one_process_driving_all_shards_over_rccl_group_calls(tmp_path, shards, sym, kernel, dtype, steps):
    """lssvm_mi355_problem_create_multi(devices = [0] * shards) with exchange = 1: Exchange::local_rccl -- ncclCommInitAll, then per matvec one
    ncclGroupStart / ncclGroupEnd around the shards' ncclAllReduce / ncclAllGather calls, each on its shard's stream (the mode behind plssvm::csvm,
    gpu_csvm.hpp:574-593)."""
    _need_stand_in()
    out = tmp_path / "local.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    pr = subprocess.run([sys.executable, os.path.join(HERE, "tools", "fake_rccl", "local_multi.py"), "--out", str(out), "--shards", str(shards), "--symmetric", str(sym),
                         "--kernel", kernel, "--dtype", dtype, "--steps", str(steps), "--points", "5000", "--features", "96"], env=env, cwd=ROOT, timeout=420)
    assert pr.returncode == 0
    r = json.load(open(out))
    assert r["stand_in_loaded"] and os.path.samefile(r["rccl_library"], STAND_IN)
    assert r["exchange"] == 1 and r["rccl_nranks"] == shards and r["rccl_rank"] == 0 and r["rccl_device"] == 0 and r["local_devices"] == shards and r["devices_used"] == shards
    assert r["symmetric"] == sym and r["iterations"] == steps and r["finite"]
    assert r["peer_exchange"] == 2 and r["peer_rccl_nranks"] == 0
    assert r["matvec_equal_bits_vs_peer"] and r["alpha_equal_bits_vs_peer"]
    if sym:
        assert r["matvec_err_vs_single"] < 64 * np.finfo(np.dtype(dtype)).eps
    else:
        assert r["matvec_equal_bits_vs_single"] and r["alpha_equal_bits_vs_single"]


def test_bench_line_of_four_ranks_reports_what_rccl_saw(tmp_path):
    """bench.py --gpus 4 --rank-devices 0,0,0,0 with the DEFAULT exchange (the library's RCCL communicator) -- through the stand-in on a one-GPU box.
    config.rccl_nranks comes from ncclCommCount, not from an option; a SCALE record of a real node can be checked against it."""
    _need_stand_in()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FAKE_RCCL_TIMEOUT_S="60")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rank-devices", "0,0,0,0", "--rccl-stand-in", STAND_IN, "--workload", "c2", "--steps", "6",
                         "--warmup", "2", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    line = json.loads(pr.stdout.strip().splitlines()[-1])
    cfg = line["config"]
    assert cfg["rccl_nranks"] == 4 and cfg["rccl_rank0_device"] == 0 and cfg["rccl_is_stand_in"] and os.path.samefile(cfg["rccl_library"], STAND_IN)
    assert cfg["shards"] == 4 and cfg["exchange"] == "RCCL all-reduce" and line["steps"] == 6 and line["n_gpus"] == 1
    assert np.isfinite(cfg["residuum_after_timed_steps"]) and line["value"] > 0 and cfg["residuum_bit_equal_on_all_ranks"] is True


def test_bench_line_of_eight_ranks_is_first_contact_ready(tmp_path):
    """What the first real SCALE record will be checked against, field by field: bench.py --gpus 8 with one process per rank and the DEFAULT exchange
    -- eight ranks on device 0 through the stand-in here -- prints ranks = 8, rccl_nranks = 8 (from ncclCommCount), the residuum as the same bits on every rank, and every
    rank's tile-kernel time per matvec.  n_gpus stays the number of DISTINCT devices (1 on this box, 8 on a node): no scaling number is made from one device."""
    _need_stand_in()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FAKE_RCCL_TIMEOUT_S="120")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rank-devices", "0,0,0,0,0,0,0,0", "--rccl-stand-in", STAND_IN, "--workload", "c2", "--steps", "6",
                         "--warmup", "2", "--no-cpu-baseline"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    line = json.loads(pr.stdout.strip().splitlines()[-1])
    cfg, roof = line["config"], line["roofline"]
    assert cfg["ranks"] == 8 and cfg["shards"] == 8 and cfg["rccl_nranks"] == 8 and cfg["rccl_is_stand_in"] and cfg["exchange"] == "RCCL all-reduce"
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["scaling"] == "strong" and line["value"] > 0 and line["setup_ms"] > 0
    assert cfg["residuum_bit_equal_on_all_ranks"] is True and np.isfinite(cfg["residuum_after_timed_steps"])
    per_rank = roof["kernel_ms_per_rank"]
    assert isinstance(per_rank, list) and len(per_rank) == 8 and all(t > 0 for t in per_rank) and roof["avg_launch_ms"] == pytest.approx(per_rank[0])


def test_bench_rebalances_the_shares_by_measured_pace(tmp_path):
    """bench.py --balance-shares (round 5): the ranks compare the tile-kernel time of their equal shares after the warm-up, set shard weights proportional to their pace
    (lssvm_mi355_set_shard_weights, the same list on every rank), rebuild their problems and run the timed steps on the new partition.  Three ranks on one device
    through the stand-in: their times differ by whatever the time sharing of the device gives, so the path may or may not trigger -- either way the line must say what
    it did, every rank must still hold the same residuum bits over the RCCL exchange, and the weights -- if set -- must be three positive numbers around one."""
    _need_stand_in()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", FAKE_RCCL_TIMEOUT_S="60")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rank-devices", "0,0,0", "--rccl-stand-in", STAND_IN, "--workload", "c2", "--steps", "6",
                         "--warmup", "4", "--no-cpu-baseline", "--balance-shares"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert pr.returncode == 0, pr.stderr[-2000:]
    cfg = json.loads(pr.stdout.strip().splitlines()[-1])["config"]
    assert cfg["rccl_nranks"] == 3 and cfg["residuum_bit_equal_on_all_ranks"] is True and np.isfinite(cfg["residuum_after_timed_steps"])
    assert isinstance(cfg["share_kernel_ms_at_equal_shares"], list) and len(cfg["share_kernel_ms_at_equal_shares"]) == 3 and min(cfg["share_kernel_ms_at_equal_shares"]) > 0
    w = cfg["shard_weights"]
    assert w is None or (len(w) == 3 and all(0.3 < x < 3.0 for x in w) and abs(sum(w) - 3.0) < 1e-3)


@pytest.mark.parametrize("world, dtype", [(3, "float64"), (4, "float32")])
def test_ranks_rebalance_their_shares_over_the_rccl_communicator(tmp_path, world, dtype):
    """lssvm_mi355_problem_rebalance with one process per rank (round 5): after four CG steps every rank gathers every rank's tile-kernel time with ONE ncclAllGather of a
    double over the library's communicator, computes the same weights and rebuilds its shard (the data, the vectors and the CG state stay); then explicit weights
    (1, 1.5, 2, ...) -- every rank must end up with the same bits again, the product after the rebalance must be the product before it up to the association, and the
    solve must go on to the float64 run's answer like the single-device solve does."""
    res = _run_ranks(tmp_path, world, ["--exchange", "1", "--kernel", "polynomial", "--dtype", dtype, "--points", "9100", "--features", "64", "--steps", "10", "--rebalance-after", "4"])
    assert len({r["alpha_sha"] for r in res}) == 1 and len({r["rho"] for r in res}) == 1 and len({r["matvec_after_rebalance_sha"] for r in res}) == 1
    assert all(r["rebalanced_by_weights"] for r in res) and len({r["rebalanced_by_measurement"] for r in res}) == 1
    eps = 2.3e-16 if dtype == "float64" else 1.2e-7
    assert all(r["matvec_after_rebalance_err"] < 256 * eps for r in res)
    r0 = [r for r in res if r["rank"] == 0][0]
    assert r0["rccl_nranks"] == world and r0["alpha_err64"] <= 2 * r0["single_err64"] + 1e-4  # (unconverged iterates of two associations: what ten CG steps amplify, 2e-6 in fp64)
