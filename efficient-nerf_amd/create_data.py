"""Pseudo-data generation, `--create_data rand` (utils/create_data.py:777-872 of the
reference): random pose + random focal -> teacher render -> [H*W, 9] = (rays_o, rays_d, rgb)
-> every `i_save` poses: concatenate, shuffle with two permutations, write `data_{k}.npy`
shards of `split_size` rays (remainder dropped) -- the on-disk format `BlenderDataset_v2`
(dataset/load_blender.py:257-324) consumes.

The teacher render is the hot part and runs on the HIP kernels (NeRFEngine).  With several
ranks the poses of a group are rendered round-robin by rank and gathered once per group
(one all-gather); rank 0 shuffles and writes, so the files are identical for any world size.

RNG: the reference draws everything from one `np.random.seed(0)` stream
(create_data.py:18): 2 draws per `get_rand_pose()` (load_blender.py:359-368), 1 per random
focal (create_data.py:816-818), two `permutation(n)` per saved group (:858-859), after
`load_blender_data` has consumed 200 `get_rand_pose()` calls (load_blender.py:89-90).
`RandStream` restates that consumption so the generated poses match the reference's.
"""
import os

import numpy as np
import torch

from . import dist as D
from .frontend import pose_spherical, to8b, write_png
from .teacher import get_rays


class RandStream:
    """The reference's single numpy RNG stream for create_data rand."""

    def __init__(self, seed=0, n_loader_poses=200):
        self.rs = np.random.RandomState(seed)  # np.random.seed(0)  (create_data.py:18)
        for _ in range(n_loader_poses):       # load_blender_data's 200 get_rand_pose() calls
            self.rand_pose()

    def rand_pose(self):
        """dataset/load_blender.py:359-368."""
        theta = -180 + self.rs.rand() * 360
        phi = -90 + self.rs.rand() * 90
        return pose_spherical(theta, phi, 4)

    def rand_focal_scale(self):
        return self.rs.rand() + 1  # focal * (np.random.rand() + 1): [1, 2)

    def permutation(self, n):
        return self.rs.permutation(n)


def create_rand(engine, H, W, focal, n_pose_kd, datadir_new, use_rand_focal=True, i_save=100, split_size=4096,
                stream=None, rm_existing_data=False, log=print, save_png=5):
    """Returns the number of `.npy` shards written by this call (rank 0; other ranks 0)."""
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_initialized() else 1
    rank = tdist.get_rank() if tdist.is_initialized() else 0
    stream = stream or RandStream()
    split = 0
    if rank == 0:
        if os.path.exists(datadir_new) and rm_existing_data:
            import shutil
            shutil.rmtree(datadir_new)
        os.makedirs(datadir_new, exist_ok=True)
        split = len([x for x in os.listdir(datadir_new) if x.endswith('.npy')])  # keep existing shards (:789-795)
    first_split = split
    dev = engine.device
    group = []  # (index in group, [H*W, 9]) rendered by this rank
    for i in range(1, n_pose_kd + 1):
        pose = stream.rand_pose()                       # every rank advances the same stream
        focal_ = focal * stream.rand_focal_scale() if use_rand_focal else focal
        if (i - 1) % world == rank:
            rays_o, rays_d = get_rays(H, W, focal_, pose[:3, :4], device=dev)  # get_rays1 (:819)
            out = engine.render_rays(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3))
            data_ = torch.cat([rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), out['rgb_map']], dim=-1)  # [H*W, 9]
            group.append(((i - 1) % i_save, data_))
            if rank == 0 and i <= save_png:
                write_png(os.path.join(datadir_new, f'pseudo_sample_{i}.png'),
                          to8b(out['rgb_map'].view(H, W, 3).cpu().numpy()))
        if i % i_save == 0:
            n_in_group = i_save
            data = _assemble_group(group, n_in_group, H * W, world, dev)
            group = []
            if rank == 0:
                # shuffle rays: data[rand_ix1][rand_ix2]  (:858-860)
                ix1 = stream.permutation(data.shape[0])
                ix2 = stream.permutation(data.shape[0])
                data = data[torch.from_numpy(ix1).to(dev)][torch.from_numpy(ix2).to(dev)].cpu().numpy()
                num = data.shape[0] // split_size * split_size
                for ix in range(0, num, split_size):
                    split += 1
                    np.save(os.path.join(datadir_new, f'data_{split}.npy'), data[ix:ix + split_size])
                log(f'[{i}/{n_pose_kd}] Saved data at "{datadir_new}"')
            else:
                stream.permutation(n_in_group * H * W)  # keep the stream in step on every rank
                stream.permutation(n_in_group * H * W)
    return split - first_split


def _assemble_group(group, n_in_group, n_rays, world, dev):
    """All poses of a save group in pose order, on every rank: [n_in_group * n_rays, 9]."""
    if world == 1:
        return torch.cat([d for _, d in sorted(group, key=lambda x: x[0])], 0)
    import torch.distributed as tdist
    per_rank = (n_in_group + world - 1) // world
    slab = torch.zeros((per_rank, n_rays, 9), dtype=torch.float32, device=dev)
    for slot, (_, d) in enumerate(sorted(group, key=lambda x: x[0])):
        slab[slot] = d
    out = torch.empty((world * per_rank, n_rays, 9), dtype=torch.float32, device=dev)
    tdist.all_gather_into_tensor(out, slab)
    out = out.view(world, per_rank, n_rays, 9)
    # pose j of the group was rendered by rank j % world in its slot j // world
    ordered = [out[j % world, j // world] for j in range(n_in_group)]
    return torch.cat(ordered, 0)


def main(argv=None):
    """`python create_data.py --create_data rand --config configs/lego.txt --teacher_ckpt X.tar
    --n_pose_kd N --datadir_kd old:new` (README.md:79 of the reference)."""
    import argparse
    from . import frontend as fe
    from . import NeRFEngine, PRECISIONS
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument('--create_data', type=str, default='spiral_evenly_spaced')
    ap.add_argument('--teacher_ckpt', type=str, default='')
    ap.add_argument('--n_pose_kd', type=int, default=100)
    ap.add_argument('--datadir_kd', type=str, default='')
    ap.add_argument('--no_rand_focal', action='store_true')
    ap.add_argument('--rm_existing_data', action='store_true')
    ap.add_argument('--create_data_chunk', type=int, default=100)
    ap.add_argument('--split_size', type=int, default=4096)
    own, rest = ap.parse_known_args(argv)
    args = fe.parse_args(rest)
    if own.create_data != 'rand':
        raise SystemExit(f'--create_data {own.create_data}: only `rand` (the README pipeline) is built')
    if ':' not in own.datadir_kd or not own.teacher_ckpt:
        raise SystemExit('need --datadir_kd old:new and --teacher_ckpt X.tar')
    rank, local_rank, world = D.init()
    torch.cuda.set_device(local_rank)
    ckpt = fe.load_checkpoint(own.teacher_ckpt)
    _, (H, W, focal) = fe.load_test_poses(args)
    prec = PRECISIONS[args.precision]
    eng = NeRFEngine(H, W, focal, 2., 6., N_samples=args.N_samples, N_importance=args.N_importance,
                     white_bkgd=args.white_bkgd, precision=prec)
    eng.load_state_dicts(ckpt['network_fn_state_dict'], ckpt['network_fine_state_dict'])
    n = create_rand(eng, H, W, focal, own.n_pose_kd, own.datadir_kd.split(':')[1], not own.no_rand_focal,
                    i_save=own.create_data_chunk, split_size=own.split_size, rm_existing_data=own.rm_existing_data,
                    log=print if rank == 0 else (lambda *a, **k: None))
    if rank == 0:
        print(f'wrote {n} shard(s) of {own.split_size} rays')
    return 0
