"""BASELINE config #5 on the HIP path: `sharding.solve_sharded` with the batched device solver -- world 1 in-process, and
a 2-rank dry run (both ranks share the one GPU of the test box; the gather goes over gloo on host copies, standing in
for the RCCL gather that needs two GPUs)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPE = (96, 130)      # (M, N)
LAM = 0.15


def _image(k):
    return np.random.default_rng(500 + k).standard_normal(SHAPE)


def _stack(a, b):
    import torch
    # (B, N, M) contiguous: image b column-major
    return torch.from_numpy(np.stack([np.ascontiguousarray(_image(k).T) for k in range(a, b)]) if b > a else np.zeros((0, SHAPE[1], SHAPE[0]))).cuda()


def test_sharded_device_solver_world1(oracle):
    from proxtv_amd import sharding
    n = 5
    local, full = sharding.solve_sharded(_stack, n, sharding.device_dr_solver(LAM), gather_to=0)
    assert full is local and tuple(local.shape) == (n, SHAPE[1], SHAPE[0])
    got = local.cpu().numpy()
    for k in range(n):
        assert_close(got[k].T, oracle.dr2(_image(k), LAM)[0], tol=1e-11, what=f"image {k}")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from proxtv_amd import sharding
    solve_gpu = sharding.device_dr_solver(LAM)
    local, full = sharding.solve_sharded(_stack, n_items, lambda x: solve_gpu(x).cpu(), gather_to=0)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_device_solver_two_ranks_one_gpu(tmp_path, oracle):
    import torch.multiprocessing as mp
    n = 5                       # ragged: 3 + 2
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    full = np.load(out)
    assert full.shape == (n, SHAPE[1], SHAPE[0])
    for k in range(n):
        assert_close(full[k].T, oracle.dr2(_image(k), LAM)[0], tol=1e-11, what=f"image {k}")
