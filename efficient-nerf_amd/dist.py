"""Multi-GPU plumbing for the ray-sharded render: one process per GPU, rows of a frame
split contiguously across ranks, ONE collective (all-gather over RCCL/xGMI; `nccl` backend
on ROCm, `gloo` in the CPU tests) to assemble the image.  The reference has no
equivalent (it renders on one GPU, main.py:473); rays are independent (main.py:90-104),
so there is no other data-path exchange."""
import os

import torch
import torch.distributed as dist


# The hosts of this pool support only dmabuf IPC: without this RCCL / device-tensor sharing across processes fails with
# `hipIpcGetMemHandle: invalid argument`.  Read by the HIP runtime when it initialises, i.e. at the first device call.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


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


class RowGather:
    """Assembles row-sharded frames on every rank with the library's one collective (include/r2l_hip.h:
    r2l_gather_image = one grouped RCCL launch over xGMI).  The frames arrive in [frame][row] order in a
    pre-allocated buffer: no copy follows the collective.  One instance per (process, device); the 128-byte RCCL id
    travels from rank 0 over the torch.distributed process group (host side, once)."""

    def __init__(self, rank, world, device, group=None):
        import ctypes as C
        from . import _lib
        self.rank, self.world, self.device = rank, world, torch.device(device)
        idbuf = (C.c_char * 128)()
        # every step below is agreed on by all ranks before anyone acts on it: a rank that failed alone would otherwise
        # fall back to torch.distributed while its peers wait inside the RCCL bootstrap.  Pre-flight: can every rank bind
        # RCCL at all?  Rank 0 makes the id (ncclGetUniqueId opens the bootstrap socket and its root thread: once, where it
        # is used); the others only bind the library (r2l_comm_available)
        err = ''
        try:
            if rank == 0:
                _lib.check(_lib.lib().r2l_comm_unique_id(C.cast(idbuf, C.c_void_p)))
            else:
                _lib.check(_lib.lib().r2l_comm_available())
        except Exception as e:
            err = str(e) or repr(e)
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
            err = next((e for e in errs if e), '')
            obj = [bytes(idbuf)]
            dist.broadcast_object_list(obj, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            idbuf = (C.c_char * 128).from_buffer_copy(obj[0])
        if err:
            raise RuntimeError(err)
        self._comm = C.c_void_p()
        try:
            with torch.cuda.device(self.device):
                _lib.check(_lib.lib().r2l_comm_create(C.byref(self._comm), rank, world, C.cast(idbuf, C.c_void_p)))
        except Exception as e:
            err = str(e) or repr(e)
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
            err = next((e for e in errs if e), '')
        if err:
            self.close()
            raise RuntimeError(err)
        self._out = None

    def gather(self, local, H, W, out=None):
        """local [F, rows_local*W, C] f32 on the device -> [F, H*W, C]: `out` when given (a contiguous device tensor the
        collective writes straight into), else the instance's own buffer, which the NEXT call overwrites -- consume or
        copy it first (frontend.render_path gathers into its frame stack)."""
        from . import _lib
        F, _, Cc = local.shape
        r0, r1 = row_shard(H, self.rank, self.world)
        assert local.shape[1] == (r1 - r0) * W and local.dtype == torch.float32 and local.is_cuda, local.shape
        if out is None:
            if self._out is None or self._out.shape != (F, H * W, Cc):
                self._out = torch.empty((F, H * W, Cc), dtype=torch.float32, device=local.device)
            out = self._out
        assert out.numel() == F * H * W * Cc and out.is_contiguous() and out.dtype == torch.float32 and out.device == local.device
        with torch.cuda.device(local.device):
            _lib.check(_lib.lib().r2l_gather_image(self._comm, _lib.dptr(local.contiguous()), _lib.dptr(out), F, H, W * Cc,
                                                   _lib.current_stream()))
        return out.view(F, H * W, Cc)

    def close(self):
        try:
            from . import _lib
        except ImportError:  # interpreter shutdown
            return
        if getattr(self, '_comm', None) is not None and self._comm.value and _lib._lib is not None:
            _lib.lib().r2l_comm_destroy(self._comm)
            self._comm = None

    __del__ = close


_row_gather = {}


_gather_used = {}


def gather_backend(device_index, world):
    """which implementation assembled the frames of this process: for bench.py's JSON line"""
    return _gather_used.get((device_index, world), 'none')


def require_rccl():
    """R2L_REQUIRE_RCCL=1: the library's own RCCL collective or an error -- never the torch.distributed stand-in
    (scaling runs: a record must say which implementation it timed, and a run that asked for this one must not time
    another)"""
    return os.environ.get('R2L_REQUIRE_RCCL', '0') not in ('', '0')


def gather_rows(local, H, W, world, group=None, force_collective=False, out=None):
    """local: [F, rows_local*W, C] slab of F frames rendered by this rank (its row shard).
    Returns [F, H*W, C] on every rank (`out`, when given: the result is written there).  Device tensors under the nccl
    backend (or world 1 with force_collective: the GPU test of the collective) go through RowGather (RCCL, frame-major
    output, no copy); the gloo path (CPU tests, rehearsals) gathers once with torch.distributed and reorders on the
    host side."""
    if world == 1 and not force_collective:
        if out is not None:
            out.view(local.shape).copy_(local)
            return out.view(local.shape)
        return local
    use_rccl = local.is_cuda and (world == 1 or dist.get_backend(group) == 'nccl')
    if use_rccl:
        key = (local.device.index, world)
        if key not in _row_gather:
            try:
                _row_gather[key] = RowGather(dist.get_rank(group) if world > 1 else 0, world, local.device, group)
            except Exception as e:  # e.g. RCCL cannot be bound by the library: the same collective through torch.distributed
                if world == 1 or require_rccl():
                    raise
                import sys
                print(f'[dist] r2l_gather_image unavailable ({e}); assembling with torch.distributed (RCCL) instead', file=sys.stderr)
                _row_gather[key] = None
        if _row_gather[key] is not None:
            _gather_used[key] = 'r2l_gather_image (RCCL, C-ABI)'
            return _row_gather[key].gather(local, H, W, out=out)
    if require_rccl() and local.is_cuda:
        raise RuntimeError('R2L_REQUIRE_RCCL=1 but the process group is %s: r2l_gather_image needs the nccl backend'
                           % dist.get_backend(group))
    F, _, Cc = local.shape
    rank = dist.get_rank(group)
    _gather_used[(local.device.index, world)] = 'torch.distributed all_gather (%s)' % (
        'RCCL' if dist.get_backend(group) == 'nccl' else dist.get_backend(group))
    sizes = [(row_shard(H, r, world)[1] - row_shard(H, r, world)[0]) * W for r in range(world)]
    assert local.shape[1] == sizes[rank], (local.shape, sizes[rank])
    mx = max(sizes)
    if local.shape[1] != mx:
        pad = torch.zeros((F, mx - local.shape[1], Cc), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 1)
    got = all_gather_cat(local, group).view(world, F, mx, Cc)
    if all(s == mx for s in sizes):
        res = got.permute(1, 0, 2, 3).reshape(F, world * mx, Cc)
    else:
        res = torch.cat([got[r, :, :sizes[r]] for r in range(world)], 1)
    if out is not None:
        out.view(res.shape).copy_(res)
        return out.view(res.shape)
    return res


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


def agree_act_exponents(eng, group=None):
    """R2L_PREC_FP16_FP8 measures its bf6 activation exponents on the first render's own rays (include/r2l_hip.h) -- under
    row sharding every rank sees other rays and would end with its own set, and the rows of one assembled frame would come
    from slightly different arithmetic.  Call once after the first render of every rank: the ranks take the element-wise
    maximum (what one GPU would have measured on all their samples together).  Every mode with calibrated operand scales takes
    part: fp16_fp8, fp16_e4m3 and the two-part modes fp16_split / fp16_split8, whose low-precision blocks read the same exponents
    (an explicit `--precision fp16_split` on several ranks kept per-rank exponents).  Synchronous; no-op for one rank or
    for the modes without scales."""
    from .r2l import SPLIT_MODES
    if not dist.is_initialized() or dist.get_world_size(group) == 1 or eng.precision not in SPLIT_MODES or eng.n_block == 0:
        return None
    ex = torch.tensor(eng.act_exponents(), dtype=torch.int32)
    if dist.get_backend(group) == 'nccl':
        ex = ex.to(eng.device)
    dist.all_reduce(ex, op=dist.ReduceOp.MAX, group=group)
    ex = [int(v) for v in ex.cpu()]
    eng.set_act_exponents(ex)
    return ex


def agree_precision(eng, group=None):
    """`--precision auto` measures its rung on every rank (R2LEngine.choose_precision / choose_split: a bisection over maxima that are
    noisy at the 1e-5 level).  The ranks render rows of the SAME frames, so they must run the same arithmetic and enter the same watch
    collectives: rank 0's (mode, split_block) is broadcast and every rank adopts it; the activation exponents follow
    (agree_act_exponents).  Returns (mode, split_block) in use.  No-op for one rank."""
    mode, split = int(eng.precision), -1 if getattr(eng, 'split_block', None) is None else int(eng.split_block)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return mode, split
    t = torch.tensor([mode, split], dtype=torch.int32)
    if dist.get_backend(group) == 'nccl':
        t = t.to(eng.device)
    dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    m0, s0 = (int(v) for v in t.cpu())
    if m0 != mode:
        eng.set_precision(m0)
    if s0 >= 0 and (m0 != mode or s0 != split):
        eng.set_split_block(s0)
    elif s0 < 0:
        eng.split_block = None
    agree_act_exponents(eng, group)
    return m0, s0


def check_ranges(eng, log=None, group=None):
    """R2LEngine.check_ranges for row-sharded runs: a rank whose rows left the calibrated range makes EVERY rank act
    -- render range-guarded ('measure'), raise its exponents to the common maximum, fall back together when `--precision
    auto`'s limit is exceeded --, so the rows of one assembled frame never come from different arithmetic.  Returns None or
    the reason to render the batch again (R2LEngine.check_ranges).  ONE small all-reduce per call when nothing trips (its
    four decision values travel together); plain eng.check_ranges for one rank."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return eng.check_ranges(log=log)

    def rank_max(vals):      # list of ints -> their element-wise maximum over the ranks
        t = torch.tensor([int(v) for v in vals], dtype=torch.int32)
        if dist.get_backend(group) == 'nccl':
            t = t.to(eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return [int(x) for x in t.cpu()]

    return eng.check_ranges(log=log, rank_max=rank_max, agree=lambda e: agree_act_exponents(e, group))


def barrier_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
