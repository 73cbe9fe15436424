"""CPU, world_size 2 over gloo: the row-block sharding of the implicit matvec.

Every rank evaluates ITS row block of K*d (here with the CPU oracle's row-owned statement of the product), the slices are
exchanged with one all-gather, and every rank must hold the same full vector as the unsharded product -- exactly the
exchange libplssvm_amd performs with ncclAllGather on the GPU box (full-square variant).  The default symmetric variant
(tiles on/below the diagonal only, row blocks dealt by equal area, mirrored column sums, one all-reduce) is restated the
same way.  Also covers the unique-id hand-off used to bootstrap the library's communicator."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plssvm_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, kernel, dtype_name, result_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    os.environ["OMP_NUM_THREADS"] = "2"
    import oracle_lib as ol

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt = np.dtype(dtype_name)
        rng = np.random.default_rng(11)  # same data on every rank (replicated X)
        N, d = 391, 9
        X = rng.uniform(-1, 1, size=(N, d)).astype(dt)
        dvec = rng.uniform(1, 2, size=N - 1).astype(dt)
        kw = dict(degree=3, gamma=1.0 / d, coef0=0.25)
        o = ol.oracle()
        q = o.q(kernel, X, **kw)
        n = N - 1
        r0, r1 = sharding.row_block_partition(n, world)[rank]
        local = o.matvec_rows(kernel, X, q, dvec, np.zeros(n, dt), 2.0, 1.0, 1.0, r0, r1, **kw)
        # all-gather of equal-sized padded slices, as the library does in place on Kv
        slice_len = sharding.padded_vector_length(n, world) // world
        mine = torch.zeros(slice_len, dtype=torch.from_numpy(local).dtype)
        mine[: r1 - r0] = torch.from_numpy(local[r0:r1])
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        full = torch.cat(gathered).numpy()[:n]
        want = o.matvec_rows(kernel, X, q, dvec, np.zeros(n, dt), 2.0, 1.0, 1.0, 0, n, **kw)
        assert np.array_equal(full, want), "sharded result must be bit-identical to the unsharded row-owned product"
        sym = o.matvec(kernel, X, q, dvec, np.zeros(n, dt), 2.0, 1.0, 1.0, **kw)
        assert ol.rel_inf(full, sym) < 64 * np.finfo(dt).eps
        # unique-id hand-off (any 128 bytes drawn on rank 0 must arrive everywhere)
        token = bytes(range(128))
        got = sharding.exchange_unique_id(dist, lambda: token)
        assert got == token
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kernel, dtype_name", [("rbf", "float32"), ("polynomial", "float64"), ("linear", "float32")])
def test_row_sharded_matvec_world2(tmp_path, kernel, dtype_name):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, kernel, dtype_name, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _sym_worker(rank, world, port, kernel, dtype_name, result_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    os.environ["OMP_NUM_THREADS"] = "2"
    import oracle_lib as ol

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dt = np.dtype(dtype_name)
        rng = np.random.default_rng(13)
        N, d = 700, 6
        X = rng.uniform(-1, 1, size=(N, d)).astype(dt)
        dvec = rng.uniform(1, 2, size=N - 1).astype(dt)
        kw = dict(degree=3, gamma=1.0 / d, coef0=0.25)
        o = ol.oracle()
        q = o.q(kernel, X, **kw)
        n = N - 1
        T = sharding.TILE
        # explicit matrix of the reduced system (column j = A-bar e_j), from the oracle's row-owned product
        A = np.empty((n, n), dt)
        for j in range(n):
            e = np.zeros(n, dt)
            e[j] = 1
            A[:, j] = o.matvec_rows(kernel, X, q, e, np.zeros(n, dt), 2.0, 1.0, 1.0, 0, n, **kw)
        b0, b1 = sharding.sym_block_partition(n, world)[rank]
        part = np.zeros(sharding.padded_vector_length(n, 1), np.float64)  # starts from zero, like Kv before the all-reduce
        for ib in range(b0, b1):
            rows = slice(ib * T, min((ib + 1) * T, n))
            for jt in range(ib + 1):
                cols = slice(jt * T, min((jt + 1) * T, n))
                blk = A[rows, cols].astype(np.float64)
                part[rows] += blk @ dvec[cols]
                if jt < ib:  # strictly below the diagonal: the mirrored tile (jt, ib) is never evaluated
                    part[cols] += blk.T @ dvec[rows]
        t = torch.from_numpy(part)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        want = A.astype(np.float64) @ dvec
        assert np.max(np.abs(t.numpy()[:n] - want)) < 1e-9 * np.max(np.abs(want))
        # every tile on or below the diagonal is owned by exactly one rank, and the areas are balanced
        alg = [sharding.work_share(n, world, r, True) for r in range(world)]
        tiles = -(-n // T)
        assert abs(sum(a for a, _ in alg) - float(n) * n) < 1e-6 * n * n
        assert sum(e for _, e in alg) == tiles * (tiles + 1) // 2 * T * T
        assert abs(sum(sharding.triangle_share(n, world, r) for r in range(world)) - n * (n + 1) / 2) < 0.5
        open(os.path.join(result_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kernel, dtype_name", [("rbf", "float64"), ("polynomial", "float64")])
def test_symmetric_sharded_matvec_world2(tmp_path, kernel, dtype_name):
    world = 2
    port = _free_port()
    mp.spawn(_sym_worker, args=(world, port, kernel, dtype_name, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("n", [1, 127, 128, 129, 390, 49_999, 999_999])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_symmetric_partition_is_a_balanced_cover(n, world):
    parts = sharding.sym_block_partition(n, world)
    tiles = -(-n // sharding.TILE)
    assert parts[0][0] == 0 and parts[-1][1] == tiles and all(a1 == b0 and a0 <= a1 for (a0, a1), (b0, _) in zip(parts, parts[1:]))
    executed = [sharding.work_share(n, world, r, True)[1] for r in range(world)]
    assert sum(executed) == tiles * (tiles + 1) // 2 * sharding.TILE ** 2
    if tiles >= 64 * world:  # enough blocks: every rank's area is within a few percent of the mean
        mean = sum(executed) / world
        assert max(abs(e - mean) for e in executed) < 0.05 * mean


@pytest.mark.parametrize("n", [1, 127, 128, 129, 390, 4095, 49_999, 999_999])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_the_librarys_own_partition_equals_the_python_restatement(n, world):
    """lssvm_mi355_shard_blocks is the function Problem<T> partitions with (plssvm_amd/csrc/lssvm_problem.hip: shard_blocks); the Python
    copy in plssvm_amd/sharding.py (used for bench.py's flop accounting and by the gloo tests above) must agree with it block for block."""
    from plssvm_amd import _capi

    sym = sharding.sym_block_partition(n, world)
    rows = sharding.row_block_partition(n, world)
    tiles = (n + sharding.TILE - 1) // sharding.TILE
    covered = 0
    for r in range(world):
        assert _capi.shard_blocks(n + 1, world, r, True) == sym[r]
        b0, b1 = _capi.shard_blocks(n + 1, world, r, False)
        assert (min(b0 * sharding.TILE, n), min(b1 * sharding.TILE, n)) == rows[r]
        covered += sym[r][1] - sym[r][0]
    assert covered == tiles and sym[0][0] == 0 and sym[-1][1] == tiles


def test_python_and_library_partitions_agree_over_a_sweep_of_shapes():
    """the even shard boundaries are computed twice -- `sym_block_boundary` (llround) in the library, `sym_block_partition` in plssvm_amd/sharding.py --
    and must match exactly (peer connection, flop accounting).  Every number of row blocks from 1 to 2 500 and every world from 1 to 16, through the host-only C entry
    point (no device needed)."""
    from plssvm_amd import _capi

    for tiles in list(range(1, 2501)) + [7813, 15626, 23438]:
        n = tiles * sharding.TILE
        for world in range(1, 17):
            sym = sharding.sym_block_partition(n, world)
            assert [_capi.shard_blocks(n + 1, world, r, True) for r in range(world)] == sym, (tiles, world)


def test_bench_spawns_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` typed as is must start two child ranks itself (no torchrun).  Without a GPU every child stops at the
    loud "needs an MI355X" check -- which proves the parent spawned ranks with RANK / WORLD_SIZE set and returned their exit code
    (tests/test_host_logic.py::test_bench_parent_stops_when_its_ranks_fail covers the sibling handling)."""
    import subprocess
    import sys

    from plssvm_amd import _capi

    if _capi.device_count() > 0:
        pytest.skip("on a GPU box this would run the full benchmark; the spawn path is exercised by bench.py itself there")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "c2"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0
    # (the parent stops the siblings of a rank that failed: the second rank may be terminated before it reaches its own check)
    assert 1 <= out.stderr.count("bench.py needs an MI355X") <= 2, out.stderr


def test_weighted_partitions_of_python_and_library_agree():
    """Round 5: shares by weight (devices of unequal pace; lssvm_mi355_set_shard_weights).  The library's partition (host-only entry point, no device needed) and its
    mirror in plssvm_amd/sharding.py must agree to the block for any weights -- both ranks' flop accounting and the peer connection rest on it --, the boundaries stay
    even and monotone, the shares cover the triangle, their areas follow the weights within one pair of row blocks, and weights of another length than the world mean
    equal shares."""
    from plssvm_amd import _capi

    rng = np.random.default_rng(17)
    try:
        for trial in range(1500):
            world = int(rng.integers(1, 17))
            tiles = int(rng.integers(1, 9000))
            n = tiles * sharding.TILE
            w = (rng.uniform(0.8, 1.2, size=world) if trial % 3 else rng.uniform(0.01, 100.0, size=world)).tolist()
            _capi.set_shard_weights(w)
            parts = sharding.sym_block_partition(n, world, w)
            assert [_capi.shard_blocks(n + 1, world, r, True) for r in range(world)] == parts, (tiles, w)
            assert parts[0][0] == 0 and parts[-1][1] == tiles and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert all(b % 2 == 0 or b == tiles for _, b in parts)
            if tiles > 64 * world:
                total = tiles * (tiles + 1) / 2
                for r, (b0, b1) in enumerate(parts):
                    area = (b1 * (b1 + 1) - b0 * (b0 + 1)) / 2
                    assert abs(area - w[r] / sum(w) * total) <= 2.5 * (tiles + 2), (tiles, w, parts)
        _capi.set_shard_weights([1.0, 2.0, 3.0])  # three weights, a world of four: equal shares
        assert [_capi.shard_blocks(1000 * 128 + 1, 4, r, True) for r in range(4)] == sharding.sym_block_partition(1000 * 128, 4)
        with pytest.raises(Exception):
            _capi.set_shard_weights([1.0, -1.0])
    finally:
        _capi.set_shard_weights(None)
    assert [_capi.shard_blocks(1000 * 128 + 1, 4, r, True) for r in range(4)] == sharding.sym_block_partition(1000 * 128, 4)
