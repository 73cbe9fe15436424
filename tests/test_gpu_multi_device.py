"""GPU (-m gpu): the row-block sharded solve THROUGH THE LIBRARY (SURVEY.md 8 rows a11 / e).

The reference drives all its devices from one process behind csvm::fit and sums their partial results in
gpu_csvm::device_reduction (include/plssvm/backends/gpu_csvm.hpp:283-299, :449-475, :574-593; tested by
tests/backends/generic_csvm_tests.hpp:495-540).  Here:

  * single process, several shards -- lssvm_mi355_solve_multi_* / lssvm_mi355_problem_create_multi.  A device ordinal may be
    listed several times, so the whole sharded path (partition, per-shard work-item lists and column slabs, zero-initialised
    partial vectors, the exchange, bit-equal CG scalars on every shard, the stop decision) runs on a ONE-GPU box with the
    peer-kernel exchange; with >= 2 devices the same tests run on distinct devices, over RCCL and over the peer kernels;
  * one process per GPU -- two fresh child processes (started before this process's children touch the GPU), each calling
    lssvm_mi355_comm_init and running rank r of 2 over RCCL (needs >= 2 devices; skipped otherwise).

Tolerances: the shards re-associate the fixed-order sums of the symmetric variant, so results agree with the single-device run
within the kernel-level bar (64 eps of the vector's scale); the full-square variant with an equal work split is BITWISE
independent of the number of shards (row-owned sums), asserted with array_equal.  CG trajectories are compared after THREE
iterations: with the reference's start vector x0 = 1 the residual falls by ten orders of magnitude in the first iterations and the
recursion then amplifies a 1e-16 re-association to 1e-3 within two more steps -- in float64, for one device against itself with another
chunking just the same (measured: delta_4 equal to 9 digits, delta_5 to 3; profiles/archive/r02_sharded_cg_sensitivity.log); converged
solves are compared at the accuracy the stop criterion defines.
"""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
from conftest import HERE, ROOT
from plssvm_amd import _capi, backend
from plssvm_amd.datagen import make_blobs_pm1
from plssvm_amd.exceptions import InvalidParameterError
from plssvm_amd.parameter import Parameter

pytestmark = pytest.mark.gpu

CASES = [("rbf", np.float32, 3000, 128), ("linear", np.float32, 2100, 200), ("polynomial", np.float64, 2500, 64), ("rbf", np.float64, 1300, 40),
         ("polynomial", np.float32, 700, 9), ("linear", np.float64, 260, 5), ("rbf", np.float32, 1700, 600)]  # (600 features: feature panels inside a tile)


def _device_lists():
    lists = [[0, 0], [0, 0, 0], [0] * 8]
    if _capi.device_count() >= 2:
        lists.append([0, 1])
    return lists


@pytest.mark.parametrize("sym", [1, 0])
@pytest.mark.parametrize("kernel, dtype, N, d", CASES)
def test_sharded_matvec_and_cg_equal_the_single_device_run(kernel, dtype, N, d, sym):
    X, y = make_blobs_pm1(N, d, seed=33, dtype=dtype)
    p = Parameter(kernel_type=kernel, cost=2.0)
    n = N - 1
    v = np.random.default_rng(9).uniform(-1, 1, size=n).astype(dtype)
    zero = np.zeros(n, dtype)
    eps = np.finfo(dtype).eps
    _capi.set_option("symmetric", sym)
    if not sym:
        _capi.set_option("j_chunk_tiles", 2)  # equal chunking on every shard count => row sums associate identically => equal bits
    with backend.ResidentProblem(p, X) as prob:
        single = prob.matvec(v, zero, 1.0)
        prob.cg_begin(y, 1e-30)
        prob.cg_step(3)
        a1, rho1, info1 = prob.cg_finish()
    assert info1["devices_used"] == 1 and info1["exchange"] == 0
    scale = np.max(np.abs(single))
    # fp32 CG amplifies the re-association of the sums from iteration to iteration: the yardstick for an fp32
    # trajectory is its distance to the same iterations in float64, which the sharded run may not exceed by more than 2x (+1e-4)
    a64 = a1
    if dtype == np.float32:
        with backend.ResidentProblem(p, X.astype(np.float64)) as prob:
            prob.cg_begin(y.astype(np.float64), 1e-30)
            prob.cg_step(3)
            a64 = prob.cg_finish()[0]
    for devices in _device_lists():
        with backend.ResidentProblem(p, X, devices=devices) as prob:
            got = prob.matvec(v, zero, 1.0)
            prob.cg_begin(y, 1e-30)
            prob.cg_step(3)
            a, rho, info = prob.cg_finish()  # also asserts (the shard check of cg_finish) that all shards hold bit-equal CG scalars
        assert info["devices_used"] == len(devices) and info["local_devices"] == len(devices) and info["symmetric"] == info1["symmetric"]
        assert info["exchange"] == (2 if len(set(devices)) < len(devices) else 1)
        if info["symmetric"]:
            assert np.max(np.abs(got - single)) < 64 * eps * scale, devices
            if dtype == np.float32:
                # (alpha_N = -sum(alpha) is left out: after three fp32 iterations from x0 = 1 the sum is a cancellation of n terms and the
                # single-device run itself is 50 % off the float64 value there)
                assert ol.rel_inf(a[:-1], a64[:-1]) < 2 * ol.rel_inf(a1[:-1], a64[:-1]) + 1e-4, devices
            else:
                assert ol.rel_inf(a, a1) < 1e-8, devices
        else:
            assert np.array_equal(got, single), devices
            assert np.array_equal(a, a1) and rho == rho1, devices


@pytest.mark.parametrize("exchange", [1, 2])
def test_two_devices_over_rccl_and_over_peer_kernels(exchange):
    if _capi.device_count() < 2:
        pytest.skip("needs two devices")
    X, y = make_blobs_pm1(6000, 128, seed=5, dtype=np.float32)
    p = Parameter(kernel_type="rbf")
    a1, rho1, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 12)
    a64, rho64, _ = backend.solve_system_of_linear_equations(p, X.astype(np.float64), y.astype(np.float64), 1e-30, 12)
    _capi.set_option("exchange", exchange)
    a2, rho2, info = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 12, devices=[0, 1])
    assert info["exchange"] == exchange and info["devices_used"] == 2
    assert ol.rel_inf(a2, a64) < 2 * ol.rel_inf(a1, a64) + 1e-4
    assert abs(float(rho2) - float(rho64)) < 2 * abs(float(rho1) - float(rho64)) + 1e-4 * max(1.0, abs(float(rho64)))


def test_sharded_solve_crosses_the_residual_refresh_and_stops_like_the_single_device_solve():
    """60 iterations (the iteration-49 refresh runs a second sharded matvec inside one step) and a converging solve whose stop
    decision is taken from shard 0's delta."""
    X, y = make_blobs_pm1(900, 24, seed=4, dtype=np.float64)
    p = Parameter(kernel_type="rbf")
    a1, rho1, i1 = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 60)
    a3, rho3, i3 = backend.solve_system_of_linear_equations(p, X, y, 1e-30, 60, devices=[0, 0, 0])
    assert i1["iterations"] == i3["iterations"] == 60 and i3["matvec_launches"] == 62
    # 60 iterations run far past convergence: both runs sit on the rounding floor of the solution
    a_conv, rho_conv, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-10, 900)
    assert ol.rel_inf(a3, a_conv) < 1e-5 and ol.rel_inf(a1, a_conv) < 1e-5 and abs(float(rho3) - float(rho1)) < 1e-5
    a1, rho1, i1 = backend.solve_system_of_linear_equations(p, X, y, 1e-8, 900)
    a2, rho2, i2 = backend.solve_system_of_linear_equations(p, X, y, 1e-8, 900, devices=[0, 0])
    assert i1["converged"] == i2["converged"] == 1 and abs(int(i1["iterations"]) - int(i2["iterations"])) <= 1
    # (the stop test delta <= eps^2 delta0 sits on a noisy plateau: where the two runs stop one iteration apart they differ by that iteration's step -- a relative
    # residual of 1e-8 bounds the solution only to cond(A) x 1e-8; seen: 1.9e-4 -- so the tight bar holds for equal counts, and both runs must agree with a solve to 1e-12)
    assert ol.rel_inf(a2, a1) < (1e-5 if int(i1["iterations"]) == int(i2["iterations"]) else 1e-3)
    a_tight, _, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-12, 900)
    assert ol.rel_inf(a1, a_tight) < 1e-3 and ol.rel_inf(a2, a_tight) < 1e-3


def test_more_shards_than_row_blocks_and_automatic_device_count():
    """Shards without any row block (260 points = 3 blocks, 8 shards) contribute zeros; num_devices = 0 picks the devices itself."""
    X, y = make_blobs_pm1(260, 5, seed=2, dtype=np.float64)
    p = Parameter(kernel_type="polynomial")
    a1, rho1, _ = backend.solve_system_of_linear_equations(p, X, y, 1e-10, 260)
    a8, rho8, i8 = backend.solve_system_of_linear_equations(p, X, y, 1e-10, 260, devices=[0] * 8)
    assert i8["devices_used"] == 8 and ol.rel_inf(a8, a1) < 1e-7
    a0, rho0, i0 = backend.solve_system_of_linear_equations(p, X, y, 1e-10, 260, num_devices=0)
    assert i0["devices_used"] == 1  # fewer than 4096 points per device: one device
    assert np.array_equal(a0, a1) and rho0 == rho1
    with pytest.raises(InvalidParameterError, match="Invalid device"):
        backend.solve_system_of_linear_equations(p, X, y, 1e-10, 260, devices=[0, 4096])
    with pytest.raises(InvalidParameterError, match="num_devices must be"):
        backend.solve_system_of_linear_equations(p, X, y, 1e-10, 260, devices=[0] * 17)


def test_options_are_snapshotted_per_problem():
    """set_option changes the defaults for problems created LATER; a live problem keeps the options it was created with."""
    X, _ = make_blobs_pm1(1500, 40, seed=8, dtype=np.float32)
    p = Parameter(kernel_type="rbf")
    v = np.linspace(1, 2, 1499).astype(np.float32)
    zero = np.zeros(1499, np.float32)
    with backend.ResidentProblem(p, X) as prob:
        before = prob.matvec(v, zero, 1.0)
        _capi.set_option("gram_mode", 0)
        _capi.set_option("symmetric", 0)
        assert prob.info()["gram_mode"] == 2 and prob.info()["symmetric"] == 1
        assert np.array_equal(prob.matvec(v, zero, 1.0), before)
    with backend.ResidentProblem(p, X) as prob:
        assert prob.info()["gram_mode"] == 0 and prob.info()["symmetric"] == 0


def _run_ranks(tmp_path, world, extra, one_device):
    """Start `world` fresh child processes of tests/tools/mp_rank.py (one rank each) and collect what they wrote."""
    port = 29000 + (os.getpid() * 7 + len(os.listdir(tmp_path)) * 13 + world) % 2000
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0" if one_device else str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "tools", "mp_rank.py"), "--out", str(tmp_path / f"rank{r}.json"), *extra], env=env, cwd=ROOT))
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


@pytest.mark.parametrize("sym", [1, 0])
def test_one_process_per_gpu_world_of_two_over_rccl(tmp_path, sym):
    """Two fresh child processes, RCCL between them: rank r creates ResidentProblem(rank=r, world=2) after lssvm_mi355_comm_init and
    runs a matvec and CG steps; rank 0 compares with its own single-GPU run (tests/tools/mp_rank.py)."""
    if _capi.device_count() < 2:
        pytest.skip("needs two devices (RCCL refuses two ranks on one device)")
    res = _run_ranks(tmp_path, 2, ["--symmetric", str(sym), "--exchange", "1"], one_device=False)
    assert res[0]["alpha_sha"] == res[1]["alpha_sha"]  # both ranks hold the same bits
    assert res[0]["matvec_err"] < 64 * np.finfo(np.float32).eps and res[0]["alpha_err64"] < 2 * res[0]["single_err64"] + 1e-4
    assert res[0]["devices_used"] == 2 and res[0]["exchange"] == 1


@pytest.mark.parametrize("world, sym, kernel, dtype, steps", [(2, 1, "rbf", "float32", 8), (2, 0, "rbf", "float32", 8), (3, 1, "polynomial", "float64", 3),
                                                              (3, 0, "linear", "float32", 8), (4, 1, "linear", "float32", 55)])
def test_one_process_per_rank_over_hip_ipc(tmp_path, world, sym, kernel, dtype, steps):
    """One process per rank THROUGH THE LIBRARY without RCCL: the ranks map each other's partial K*v with HIP IPC and sum / gather them
    with the peer kernel (lssvm_mi355_problem_ipc_export / _connect).  Runs with all ranks on one device too (this pool's boxes have
    one), and on one device per rank where there are enough.  Every rank must end with the same bits; the full-square variant must
    reproduce the single-GPU bits (row-owned sums).  (fp64: three iterations -- the CG recursion amplifies the
    re-association of the sharded sums from iteration to iteration; the 55-step case crosses the residual refresh.)"""
    one_device = _capi.device_count() < world
    res = _run_ranks(tmp_path, world, ["--symmetric", str(sym), "--exchange", "2", "--kernel", kernel, "--dtype", dtype, "--points", "5000", "--features", "96", "--steps", str(steps)],
                     one_device=one_device)
    assert len({r["alpha_sha"] for r in res}) == 1 and len({r["rho"] for r in res}) == 1
    assert all(r["devices_used"] == world and r["exchange"] == 2 and r["symmetric"] == sym for r in res)
    eps = np.finfo(np.dtype(dtype)).eps
    if sym:
        assert res[0]["matvec_err"] < 64 * eps
        assert res[0]["alpha_err64"] < 2 * res[0]["single_err64"] + (1e-4 if dtype == "float32" else 1e-8)
    else:
        assert res[0]["matvec_equal_bits"] and res[0]["alpha_equal_bits"]
