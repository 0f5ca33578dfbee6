"""Multi-GPU plumbing for the ray-sharded render: one process per GPU, rows of a frame
split contiguously across ranks, ONE collective (all-gather over RCCL/xGMI; `nccl` backend
on ROCm, `gloo` in the CPU tests) to assemble the image.  The reference has no
equivalent (it renders on one GPU, main.py:473); rays are independent (main.py:90-104),
so there is no other data-path exchange."""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for world 1)."""
    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:  # R2L_DIST_BACKEND=gloo: rehearsal of an N-rank run on fewer GPUs (see local_device)
            backend = os.environ.get('R2L_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local_device(local_rank))
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


def local_device(local_rank):
    """CUDA device index of a rank: LOCAL_RANK, wrapped when a gloo rehearsal runs more ranks than GPUs."""
    n = torch.cuda.device_count()
    return local_rank % n if n else 0


def row_shard(H, rank, world):
    """Contiguous rows [r0, r1) of rank `rank`; the first H % world ranks get one more."""
    base, rem = divmod(H, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def gather_rows(local, H, W, world, group=None):
    """local: [F, rows_local*W, C] slab of F frames rendered by this rank (its row shard).
    Returns [F, H*W, C] on every rank.  Equal shards: one all_gather_into_tensor; ragged
    shards (H % world != 0): pad to the largest shard, gather once, strip."""
    if world == 1:
        return local
    F, _, Cc = local.shape
    rank = dist.get_rank(group)
    sizes = [(row_shard(H, r, world)[1] - row_shard(H, r, world)[0]) * W for r in range(world)]
    assert local.shape[1] == sizes[rank], (local.shape, sizes[rank])
    mx = max(sizes)
    if local.shape[1] != mx:
        pad = torch.zeros((F, mx - local.shape[1], Cc), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 1)
    out = torch.empty((world * F, mx, Cc), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == 'gloo':  # rehearsal only: gloo gathers through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)  # concatenation along dim 0
    out = out.view(world, F, mx, Cc)
    if all(s == mx for s in sizes):
        return out.permute(1, 0, 2, 3).reshape(F, world * mx, Cc)
    return torch.cat([out[r, :, :sizes[r]] for r in range(world)], 1)


def all_gather_cat(t, group=None):
    """[world * n, ...] = every rank's `t` [n, ...] concatenated along dim 0, on every rank: ONE all-gather
    (RCCL over xGMI; a gloo rehearsal with device tensors goes through the host)."""
    world = dist.get_world_size(group)
    t = t.contiguous()
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, t.cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, t, group=group)
    return out


def barrier_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
