"""CPU, world_size 2 over gloo: the row-block sharding of the implicit matvec.

Every rank evaluates ITS row block of K*d (here with the CPU oracle's row-owned statement of the product), the slices are
exchanged with one all-gather, and every rank must hold the same full vector as the unsharded product -- exactly the
exchange libplssvm_amd performs with ncclAllGather on the GPU box.  Also covers the unique-id hand-off used to bootstrap
the library's communicator."""

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
