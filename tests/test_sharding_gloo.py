"""The N > 1 path on CPU: world_size-2 (and 3) gloo runs of proxtv_amd.sharding with the CPU oracle standing in for
the per-GPU solver.  Checks the block partition (including ragged shards and more ranks than images) and that the one
gather returns the batch in order, identical to solving the images one by one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cpu
    from proxtv_amd import sharding
    orc = cpu.oracle()
    images = np.stack([np.random.default_rng(100 + k).standard_normal((12, 17)) for k in range(n_items)]) if n_items else \
        np.zeros((0, 12, 17))

    def get_images(a, b):
        return torch.from_numpy(images[a:b].copy())

    def solve(x):
        return torch.from_numpy(np.stack([orc.dr2(im, 0.2)[0] for im in x.numpy()]) if x.shape[0] else np.zeros((0, 12, 17)))

    # what rank 0 may allocate for the gather: the (n_items, ...) result and nothing else -- no list of per-rank buffers, no padding,
    # no concatenation (16 GiB instead of 2 x 16 GiB + a copy at BASELINE config #5)
    allocated = []
    real_alloc = sharding._alloc

    def counting_alloc(shape, like):
        t = real_alloc(shape, like)
        allocated.append(t.numel() * t.element_size())
        return t

    def no_cat(*a, **k):
        raise AssertionError("solve_sharded must not concatenate")

    sharding._alloc = counting_alloc
    real_cat, torch.cat = torch.cat, no_cat
    try:
        local, full = sharding.solve_sharded(get_images, n_items, solve, gather_to=0)
    finally:
        torch.cat = real_cat
        sharding._alloc = real_alloc
    assert allocated == ([n_items * 12 * 17 * 8] if rank == 0 else []), allocated
    a, b = sharding.shard_bounds(n_items, world, rank)
    assert local.shape[0] == b - a
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 6), (2, 5), (3, 7), (2, 1)])
def test_sharded_batch_equals_loop(tmp_path, oracle, world, n_items):
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_items, out), nprocs=world, join=True)
    full = np.load(out)
    want = np.stack([oracle.dr2(np.random.default_rng(100 + k).standard_normal((12, 17)), 0.2)[0] for k in range(n_items)])
    np.testing.assert_array_equal(full, want)


def test_shard_bounds_cover_exactly():
    from proxtv_amd import sharding
    for n in (0, 1, 7, 8, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
            assert sizes == sharding.shard_sizes(n, world)
    assert sharding.shard_bounds(512, 8, 3) == (192, 256)     # BASELINE config #5: 64 images per GPU
