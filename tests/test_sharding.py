"""Multi-GPU plumbing on CPU: world_size-2 gloo processes shard a batch, run the LOCAL frames (the oracle stands in
for the GPU engine here -- tests may use it as the checker), all-gather the heads and must reproduce the full-batch
result in order.  Covers even, uneven and tiny batches."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def test_shard_range_partitions_the_batch():
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    for n in (0, 1, 2, 7, 8, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            r = [sh.shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
    assert sh.shard_range(32768, 3, 8) == (3 * 4096, 4 * 4096)      # BASELINE config 3: 4096 frames per rank


def _worker(rank, world, port, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    from oracle.oracle import Oracle
    x = np.random.default_rng(123).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
    a, b = sh.shard_range(n, rank, world)
    local = torch.from_numpy(Oracle().run(x[a:b]))
    full = sh.all_gather_heads(local, n)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [8, 7, 1])
def test_two_rank_gloo_all_gather_reproduces_full_batch(n, oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x = np.random.default_rng(123).integers(-128, 128, (n, 56, 56, 3), dtype=np.int8)
    assert np.array_equal(got, oracle.run(x))


def test_c_shard_range_equals_the_python_one(yf):
    """yf_network_shard_range is what a C host application uses (multi-GPU); same split as sharding.py."""
    import ctypes
    lib = yf.load()
    sh = importlib.import_module("stm32h7-yolo_amd.sharding")
    for n in (0, 1, 7, 8, 4096, 32768, 32771):
        for world in (1, 2, 3, 8):
            for r in range(world):
                a, b = ctypes.c_long(), ctypes.c_long()
                lib.yf_network_shard_range(n, r, world, ctypes.byref(a), ctypes.byref(b))
                assert (a.value, b.value) == sh.shard_range(n, r, world)
