"""Batch sharding across the GPUs of a node: one process per GPU, independent images, no collective in the data path.

The reference has no batch or multi-device concept; images of a batch are independent TV problems, so the
batch is block-partitioned over ranks, every rank runs the single-GPU HIP path on its shard (all images of the shard
advance together, one launch per sweep), and a single gather over RCCL (xGMI) collects the results only if the
caller wants them in one place.  Within one image nothing is sharded: every sweep needs all fibres of the previous,
orthogonal sweep, which would be an all-to-all of the whole image 72 times per solve.

`torch.distributed` is plumbing (backend "nccl" is RCCL on ROCm; "gloo" drives the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, world, rank):
    """Block partition of range(n_items): the first n_items % world ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world):
    return [shard_bounds(n_items, world, r)[1] - shard_bounds(n_items, world, r)[0] for r in range(world)]


def device_dr_solver(lam, max_iters=0):
    """The single-GPU solver for `solve_sharded`: 2-D TV-L1 Douglas-Rachford on a stack of images held as a contiguous
    (B, N, M) CUDA float64 tensor -- image b stored column-major (M, N), which is what the library wants: the stack is
    the column-major (M, N, B) array of `proxtv_DR2_TV_batch_dev` seen from the other end.  All images of the shard
    advance together, one launch per sweep."""
    from . import device

    def solve(images):
        if images.shape[0] == 0:
            return images.clone()
        x = images.contiguous().permute(2, 1, 0)          # (M, N, B), dimension 0 fastest
        y, _ = device.tv1_2d_batch(x, lam, max_iters=max_iters)
        return y.permute(2, 1, 0)                          # back to (B, N, M), contiguous
    return solve


def _alloc(shape, like):
    """Every buffer solve_sharded allocates comes from here (the gloo test counts the bytes)."""
    return torch.empty(shape, dtype=like.dtype, device=like.device)


def solve_sharded(get_images, n_items, solve, gather_to=0, group=None):
    """Solve a batch of `n_items` independent images across the ranks of `group`.

    get_images(start, stop) -> tensor of this rank's images, shape (stop - start, ...), already on the rank's device
    solve(images)           -> tensor of the same shape (the single-GPU path)
    gather_to               -> rank that receives the full result in batch order (None: leave results sharded)

    Returns (local_result, gathered_or_None).  The only communication is the final gather, and the receiving rank allocates
    exactly ONE buffer for it -- the (n_items, ...) result itself: every shard lands in its own rows (views of that tensor),
    nothing is padded, listed or concatenated.  At BASELINE config #5 that is 16 GiB on rank 0 instead of 2 x 16 GiB plus a copy.
    Equal shards go through one `dist.gather` (RCCL: grouped send / receive over xGMI); ragged shards, which a gather of
    equal-sized tensors cannot express without padding, through the same sends and receives posted directly.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    start, stop = shard_bounds(n_items, world, rank)
    local = solve(get_images(start, stop))
    if gather_to is None or world == 1:
        return local, (local if (gather_to is not None) else None)
    local = local.contiguous()
    sizes = shard_sizes(n_items, world)
    item_shape = tuple(local.shape[1:])
    even = min(sizes) == max(sizes)
    if rank == gather_to:
        full = _alloc((int(n_items),) + item_shape, local)
        if even:
            dist.gather(local, gather_list=list(full.view((world, sizes[0]) + item_shape).unbind(0)), dst=gather_to, group=group)
            return local, full
        ops = []
        for r in range(world):
            a, b = shard_bounds(n_items, world, r)
            if r == rank:
                full[a:b].copy_(local)
            elif b > a:
                ops.append(dist.P2POp(dist.irecv, full[a:b], r, group))
        for req in (dist.batch_isend_irecv(ops) if ops else []):
            req.wait()
        return local, full
    if even:
        dist.gather(local, gather_list=None, dst=gather_to, group=group)
    elif local.shape[0] > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, gather_to, group)]):
            req.wait()
    return local, None
