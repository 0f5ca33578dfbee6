"""Pseudo-data generation, `--create_data rand` (utils/create_data.py:777-872 of the
reference): random pose + random focal -> teacher render -> [H*W, 9] = (rays_o, rays_d, rgb)
-> every `i_save` poses: concatenate, shuffle with two permutations, write `data_{k}.npy`
shards of `split_size` rays (remainder dropped) -- the on-disk format `BlenderDataset_v2`
(dataset/load_blender.py:257-324) consumes.

The teacher render is the hot part and runs on the HIP kernels (NeRFEngine).  With several
ranks the poses of a group are rendered round-robin by rank (by their index inside the group) and
gathered once per group (one all-gather); every rank then writes the shards k = rank (mod world), so
the directory is byte-identical for any world size (tests/test_create_data_cpu.py).

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
                stream=None, rm_existing_data=False, log=print, save_png=5, get_rays_fn=None):
    """Returns the number of `.npy` shards of this call (the same on every rank).

    Multi-rank (torch.distributed initialised): pose j of a save group (j = index INSIDE the group) is rendered
    by rank j % world; one all-gather per group puts the whole group on every rank; every rank applies the two
    permutations (all ranks draw the same numpy stream) and writes the shards k with k % world == rank, so the
    directory is byte-identical for any world size."""
    import torch.distributed as tdist
    world = tdist.get_world_size() if tdist.is_initialized() else 1
    rank = tdist.get_rank() if tdist.is_initialized() else 0
    get_rays_fn = get_rays_fn or get_rays
    stream = stream or RandStream()
    if rank == 0:
        if os.path.exists(datadir_new) and rm_existing_data:
            import shutil
            shutil.rmtree(datadir_new)
        os.makedirs(datadir_new, exist_ok=True)
    if world > 1:
        tdist.barrier()  # the directory exists (and is emptied) before anybody lists or writes it
    split = len([x for x in os.listdir(datadir_new) if x.endswith('.npy')])  # keep existing shards (:789-795)
    if world > 1:
        tdist.barrier()  # everybody has counted before anybody writes
    first_split = split
    dev = engine.device
    group = []  # (index in group, [H*W, 9]) rendered by this rank
    for i in range(1, n_pose_kd + 1):
        pose = stream.rand_pose()                       # every rank advances the same stream
        focal_ = focal * stream.rand_focal_scale() if use_rand_focal else focal
        j = (i - 1) % i_save                            # index inside the save group
        if j % world == rank:
            rays_o, rays_d = get_rays_fn(H, W, focal_, pose[:3, :4], device=dev)  # get_rays1 (:819)
            out = engine.render_rays(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3))
            data_ = torch.cat([rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), out['rgb_map']], dim=-1)  # [H*W, 9]
            group.append((j, data_))
            if i <= save_png:
                write_png(os.path.join(datadir_new, f'pseudo_sample_{i}.png'),
                          to8b(out['rgb_map'].view(H, W, 3).cpu().numpy()))
        if i % i_save == 0:
            data = _assemble_group(group, i_save, H * W, world, dev)
            group = []
            # shuffle rays: data[rand_ix1][rand_ix2] == data[rand_ix1[rand_ix2]]  (:858-860)
            ix1 = stream.permutation(data.shape[0])
            ix2 = stream.permutation(data.shape[0])
            comb = torch.from_numpy(ix1[ix2])
            num = data.shape[0] // split_size * split_size
            for ix in range(0, num, split_size):
                split += 1
                if split % world == rank:
                    rows = data[comb[ix:ix + split_size].to(dev)].cpu().numpy()
                    np.save(os.path.join(datadir_new, f'data_{split}.npy'), rows)
            log(f'[{i}/{n_pose_kd}] Saved data at "{datadir_new}"')
    if world > 1:
        tdist.barrier()  # all shards are on disk when any rank returns
    return split - first_split


def _assemble_group(group, n_in_group, n_rays, world, dev):
    """All poses of a save group in pose order, on every rank: [n_in_group * n_rays, 9].
    `group`: this rank's (index in group j, data) with j % world == rank."""
    group = sorted(group, key=lambda x: x[0])
    if world == 1:
        assert [j for j, _ in group] == list(range(n_in_group))
        return torch.cat([d for _, d in group], 0)
    per_rank = (n_in_group + world - 1) // world
    slab = torch.zeros((per_rank, n_rays, 9), dtype=torch.float32, device=dev)
    for j, d in group:
        slab[j // world] = d   # pose j sits on rank j % world in slot j // world
    out = D.all_gather_cat(slab).view(world, per_rank, n_rays, 9)
    return torch.cat([out[j % world, j // world] for j in range(n_in_group)], 0)


class BlenderDataset_v2:
    """Reader of the shard directory `create_rand` writes, with the indexing of the reference's
    `BlenderDataset_v2` (dataset/load_blender.py:257-324): item k = file k of the listing -> (rays_o [n,3],
    rays_d [n,dim_dir], rgb [n,dim_rgb]); `train_*.npy` files are the original data, the rest pseudo data;
    pseudo_ratio / hold_ratio sub-sample the file list with np.random.choice exactly as the reference does."""

    def __init__(self, datadir, dim_dir=3, dim_rgb=3, rand_crop_size=-1, img_H=0, img_W=0, hold_ratio=0, pseudo_ratio=1.):
        self.datadir = datadir
        names = os.listdir(datadir)
        pseudo = [f'{datadir}/{x}' for x in names if x.endswith('.npy') and not x.startswith('train_')]
        original = [f'{datadir}/{x}' for x in names if x.endswith('.npy') and x.startswith('train_')]
        assert 0 <= pseudo_ratio <= 1 or pseudo_ratio == -1
        if pseudo_ratio == -1 or (pseudo_ratio == 1 and not original):  # use all the data
            all_splits = pseudo + original
        else:
            if pseudo_ratio == 1:
                raise ValueError('pseudo_ratio = 1 with original data present divides by zero in the reference '
                                 '(load_blender.py:288-289); pass -1 to use everything')
            num_pseudo = int(len(original) / (1. - pseudo_ratio)) - len(original)
            pseudo = np.random.choice(pseudo, num_pseudo).tolist()
            all_splits = pseudo + original
        assert 0 <= hold_ratio < 1
        if hold_ratio > 0:
            all_splits = np.random.choice(all_splits, int(len(all_splits) * (1 - hold_ratio)))
        self.all_splits = all_splits
        self.dim_dir, self.dim_rgb = dim_dir, dim_rgb
        self.rand_crop_size, self.img_H, self.img_W = rand_crop_size, img_H, img_W

    def _square_rand_bbox(self):
        bbx1 = np.random.randint(0, self.img_W - self.rand_crop_size + 1)
        bby1 = np.random.randint(0, self.img_H - self.rand_crop_size + 1)
        return bbx1, bby1, bbx1 + self.rand_crop_size, bby1 + self.rand_crop_size

    def __getitem__(self, index):
        d = torch.from_numpy(np.load(self.all_splits[index])).float()  # [H, W, 9] or [n_ray, 9]
        if self.rand_crop_size > 0:
            bbx1, bby1, bbx2, bby2 = self._square_rand_bbox()
            d = d[bby1:bby2, bbx1:bbx2, :]
        return d[..., :3], d[..., 3:3 + self.dim_dir], d[..., 3 + self.dim_dir:3 + self.dim_dir + self.dim_rgb]

    def __len__(self):
        return len(self.all_splits)


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
    torch.cuda.set_device(D.local_device(local_rank))
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
