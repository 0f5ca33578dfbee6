"""Thin front-end keeping the reference's command line, `configs/*.txt` format and `.tar`
checkpoint schema for the render-only path:

    python main.py --model_name R2L --config configs/lego_noview.txt --n_sample_per_ray 16 \\
        --netwidth 256 --netdepth 88 --use_residual --trial.ON --trial.body_arch resmlp \\
        --pretrained_ckpt R2L_Blender_Models/lego.tar --render_only --render_test --testskip 1
    python main.py --model_name nerf --config configs/lego.txt --pretrained_ckpt NeRF_Blender_Models/lego.tar \\
        --render_only --render_test --testskip 1

Reference pieces restated (file:line in MingSun-Tse/Efficient-NeRF):
  flags / config file      option.py:6-358  (ConfigArgParse-style `key = value`, `#` comments)
  transforms_*.json rule   dataset/load_blender.py:39-82, 106-109
  checkpoint schema        main.py:1516-1542 (save_ckpt), main.py:482-502 (load)
  render_path loop         main.py:189-398 (R2L branch :285-325, nerf branch :276-283)
  PSNR                     utils/run_nerf_raybased_helpers.py:19-20

Everything computational goes through the HIP library; this module is plumbing.  With
several processes (torchrun) the rows of each frame are sharded across ranks and assembled
with one all-gather (dist.py).  Images on disk (PNG GT, `imageio` writes) are a next-row
item: without `imageio` GT images are not read and renders are saved as .npy / PNG.
"""
import argparse
import io
import json
import math
import os
import pickle
import struct
import sys
import time
import types
import zlib

import numpy as np
import torch

# ----------------------------------------------------------------------------------------
# flags (subset of option.py that the render-only hot path reads; same names, same defaults)
# ----------------------------------------------------------------------------------------
_BOOL = 'store_true'
FLAGS = [
    ('--config', dict(type=str, default=None)), ('--expname', dict(type=str, default=None)),
    ('--basedir', dict(type=str, default='./logs/')), ('--datadir', dict(type=str, default='./data/llff/fern')),
    ('--netdepth', dict(type=int, default=8)), ('--netwidth', dict(type=int, default=256)),
    ('--netdepth_fine', dict(type=int, default=8)), ('--netwidth_fine', dict(type=int, default=256)),
    ('--chunk', dict(type=int, default=1024 * 32)), ('--netchunk', dict(type=int, default=1024 * 64)),
    ('--N_samples', dict(type=int, default=64)), ('--N_importance', dict(type=int, default=0)),
    ('--perturb', dict(type=float, default=1.)), ('--perturb_test', dict(type=float, default=0.)),
    ('--use_viewdirs', dict(action=_BOOL)), ('--i_embed', dict(type=int, default=0)),
    ('--multires', dict(type=int, default=10)), ('--multires_views', dict(type=int, default=4)),
    ('--raw_noise_std', dict(type=float, default=0.)), ('--render_only', dict(action=_BOOL)),
    ('--render_test', dict(action=_BOOL)), ('--render_factor', dict(type=int, default=0)),
    ('--dataset_type', dict(type=str, default='llff')), ('--testskip', dict(type=int, default=8)),
    ('--white_bkgd', dict(action=_BOOL)), ('--half_res', dict(action=_BOOL)), ('--no_ndc', dict(action=_BOOL)),
    ('--lindisp', dict(action=_BOOL)), ('--model_name', dict(type=str, default='nerf')),
    ('--n_sample_per_ray', dict(type=int, default=192)), ('--pretrained_ckpt', dict(type=str, default='')),
    ('--n_pose_video', dict(type=str, default='20,4,1')), ('--video_tag', dict(type=str, default='')),
    ('--use_residual', dict(action=_BOOL)), ('--linear_tail', dict(action=_BOOL)),
    ('--layerwise_netwidths', dict(type=str, default='')), ('--act', dict(type=str, default='relu')),
    ('--focal_scale', dict(type=float, default=1.)), ('--given_render_path_rays', dict(type=str, default='')),
    ('--learn_depth', dict(action=_BOOL)), ('--plucker', dict(action=_BOOL)), ('--benchmark', dict(action=_BOOL)),
    ('--trial.ON', dict(action=_BOOL)), ('--trial.body_arch', dict(type=str, default='mlp')),
    ('--trial.res_scale', dict(type=float, default=1.)), ('--trial.n_learnable', dict(type=int, default=2)),
    ('--trial.inact', dict(type=str, default='relu')), ('--trial.outact', dict(type=str, default='none')),
    ('--trial.n_block', dict(type=int, default=-1)), ('--trial.near', dict(type=float, default=-1)),
    ('--trial.far', dict(type=float, default=-1)),
    # accepted and ignored (logging / training plumbing of the reference command lines)
    ('--project', dict(type=str, default='')), ('--screen', dict(action=_BOOL)), ('--cache_ignore', dict(type=str, default='')),
    ('--debug', dict(action=_BOOL)), ('--no_batching', dict(action=_BOOL)), ('--lrate_decay', dict(type=int, default=250)),
    ('--N_rand', dict(type=int, default=4096)), ('--precrop_iters', dict(type=int, default=0)),
    ('--precrop_frac', dict(type=float, default=.5)), ('--no_reload', dict(action=_BOOL)),
    # this front-end's own knobs
    # auto (default): fp16_fp8 (fp16 MFMA pass + bf6 correction terms, 1.7x the speed) when the checkpoint's own activation
    # ranges, measured on every ray of the first frame and watched on every frame after it, keep it inside the 1e-4 rgb
    # contract, fp16x3_asm otherwise (R2LEngine.choose_precision / check_ranges); the teacher measures fp16x1 (its layer chain as one
    # fp16 pass), then fp16_fp8, against fp16x3 (NeRFEngine.choose_precision); fp32 = the generic layer path for any network shape
    ('--precision', dict(type=str, default='auto', choices=['fp16x3', 'fp16x1', 'fp16_fp8', 'fp16_e4m3', 'fp16x3_asm', 'fp16_split', 'fp16_split8', 'fp16_mix', 'fp32', 'auto'])),
    # --precision fp16_split / fp16_split8 taken literally: how many leading ResMLP blocks run in three fp16 passes (-1: half of them; `auto`
    # measures it, and which of the two formats behind the split is cheaper)
    ('--split_block', dict(type=int, default=-1)),
    ('--synthetic_poses', dict(type=int, default=0)), ('--outdir', dict(type=str, default='')),
    ('--H', dict(type=int, default=0)), ('--W', dict(type=int, default=0)),
    # frames rendered per launch / collective / range check / host sync (0: the world size, i.e. one frame on one GPU)
    ('--frames_per_batch', dict(type=int, default=0)),
    # the watches behind `--precision auto` (render_path): the split rungs / the teacher's fast modes every watch_every-th batch / frame, the
    # whole-network rungs every 2 x watch_every-th batch; 0 switches them off (A/B of their cost: tools/cli_soak.py)
    ('--watch_every', dict(type=int, default=8)),
    # N > 1 without torchrun: main.py / create_data.py start the N ranks themselves before importing torch (launch.py)
    ('--gpus', dict(type=int, default=0)), ('--launch_timeout', dict(type=float, default=0.)),
]


def parse_config_file(path):
    """`key = value` lines, `#` comments (also trailing), blank lines; booleans as
    True/False (configs/lego_noview.txt:6-19).  Returns an ordered list of (key, value)."""
    items = []
    with open(path) as f:
        for ln, line in enumerate(f, 1):
            line = line.split('#', 1)[0].strip()
            if not line:
                continue
            if '=' not in line:
                raise ValueError(f'{path}:{ln}: expected `key = value`, got {line!r}')
            k, v = (x.strip() for x in line.split('=', 1))
            items.append((k, v))
    return items


def build_parser():
    p = argparse.ArgumentParser(prog='main.py', description=__doc__.split('\n\n')[0])
    for name, kw in FLAGS:
        p.add_argument(name, **kw)
    return p


def parse_args(argv=None):
    """Command line wins over the config file, as with ConfigArgParse (option.py:6).  Dotted
    flags become a nested namespace (`args.trial.body_arch`), option.py:335-358, 386."""
    argv = list(sys.argv[1:] if argv is None else argv)
    p = build_parser()
    pre, _ = p.parse_known_args(argv)
    cfg_argv = []
    if pre.config:
        known = {n for n, _ in FLAGS}
        for k, v in parse_config_file(pre.config):
            flag = '--' + k
            if flag not in known:
                continue  # training-only keys of the reference's configs
            kw = dict(FLAGS)[flag]
            if kw.get('action') == _BOOL:
                if v.lower() in ('true', '1', 'yes'):
                    cfg_argv.append(flag)
            else:
                cfg_argv += [flag, v]
    args = p.parse_args(cfg_argv + argv)
    trial = types.SimpleNamespace()
    for k in list(vars(args)):
        if k.startswith('trial.'):
            setattr(trial, k.split('.', 1)[1], getattr(args, k))
            delattr(args, k)
    args.trial = trial
    return args


# ----------------------------------------------------------------------------------------
# checkpoint (.tar) reader
# ----------------------------------------------------------------------------------------
class _Stub:
    """Stands in for classes of the reference that are pickled into checkpoints but not
    importable here (model.nerf_raybased.NeRF_v3_2 / ResMLP, utils.EmptyClass, smilelogging
    namespaces ...).  Keeps whatever state the pickle carries."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state if isinstance(state, dict) else {'_state': state})

    def __call__(self, *a, **k):  # pickled lambdas / functions reduce to calls
        return _Stub()


class _TolerantUnpickler(pickle.Unpickler):
    _FOREIGN = ('model', 'utils', 'smilelogging', 'option', '__main__', 'dataset', 'main')

    def find_class(self, module, name):
        root = module.split('.', 1)[0]
        if root in self._FOREIGN:
            return type(name, (_Stub,), {'__module__': module})
        return super().find_class(module, name)


_pickle_mod = types.ModuleType('r2l_tolerant_pickle')
_pickle_mod.Unpickler = _TolerantUnpickler
_pickle_mod.load = lambda f, **kw: _TolerantUnpickler(f, **kw).load()
_pickle_mod.__dict__.update({k: getattr(pickle, k) for k in ('dump', 'dumps', 'loads', 'Pickler', 'HIGHEST_PROTOCOL')})


def strip_module_prefix(sd):
    """utils/run_nerf_raybased_helpers.py:408-425 (undataparallel)."""
    return {(k.split('module.', 1)[1] if k.startswith('module.') else k): v for k, v in sd.items()}


def load_checkpoint(path):
    """torch.load of a reference `.tar` with map_location='cpu' (the reference relies on the
    saving device being present, main.py:483).  Returns the dict; `*_state_dict` entries are
    plain tensors with `module.` prefixes removed; the pickled `network_fn` object (R2L
    checkpoints, main.py:1534-1536) is tolerated and only mined for its state if the
    state_dict entry is missing."""
    ckpt = torch.load(path, map_location='cpu', pickle_module=_pickle_mod, weights_only=False)
    if not isinstance(ckpt, dict):
        raise ValueError(f'{path}: expected a dict checkpoint, got {type(ckpt).__name__}')
    for k in list(ckpt):
        if k.endswith('_state_dict') and k != 'optimizer_state_dict' and isinstance(ckpt[k], dict):
            ckpt[k] = strip_module_prefix(ckpt[k])
    if 'network_fn_state_dict' not in ckpt:
        raise KeyError(f"{path}: no 'network_fn_state_dict' (keys: {sorted(ckpt)})")
    return ckpt


def save_checkpoint(path, network_fn_state_dict, network_fine_state_dict=None, global_step=0):
    """Writes the reference's schema (main.py:1516-1542) without the pickled module."""
    to_save = {'global_step': global_step, 'best_psnr': 0, 'best_psnr_step': 0,
               'network_fn_state_dict': dict(network_fn_state_dict), 'optimizer_state_dict': {}}
    if network_fine_state_dict is not None:
        to_save['network_fine_state_dict'] = dict(network_fine_state_dict)
    torch.save(to_save, path)
    return path


# ----------------------------------------------------------------------------------------
# poses / camera
# ----------------------------------------------------------------------------------------
def pose_spherical(theta, phi, radius):
    """dataset/load_blender.py:10-28."""
    t = torch.Tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]]).float()
    ph, th = phi / 180. * np.pi, theta / 180. * np.pi
    rp = torch.Tensor([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0],
                       [0, 0, 0, 1]]).float()
    rt = torch.Tensor([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0],
                       [0, 0, 0, 1]]).float()
    return torch.Tensor([[-1, 0, 0, 0], [0, 0, 1, 0], [0, 1, 0, 0], [0, 0, 0, 1]]) @ (rt @ (rp @ t))


LEGO_CAMERA_ANGLE_X = 0.6911112070083618


def load_test_poses(args):
    """Test-split poses + intrinsics.  With a Blender datadir: transforms_test.json
    (load_blender.py:39-82: frames[::testskip], focal = .5*W/tan(.5*camera_angle_x), half_res
    halves H, W, focal :106-109).  Without data (offline): `--synthetic_poses N` evenly spaced
    pose_spherical(theta, -30, 4) views with the lego intrinsics."""
    tf = os.path.join(args.datadir, 'transforms_test.json')
    if args.synthetic_poses <= 0 and os.path.exists(tf):
        with open(tf) as fp:
            meta = json.load(fp)
        if args.render_test:
            skip = 1 if args.testskip == 0 else args.testskip
            frames = meta['frames'][::skip]
            poses = torch.tensor(np.array([f['transform_matrix'] for f in frames]).astype(np.float32))
        else:  # the video path: n_pose = 40 views on the -30 degree circle (load_blender.py:35, 91-93)
            poses = torch.stack([pose_spherical(t, -30., 4.) for t in np.linspace(-180, 180, 40 + 1)[:-1]], 0)
        H = W = args.H or 800
        angle = float(meta['camera_angle_x'])
    else:
        n = args.synthetic_poses or 4
        thetas = np.linspace(-180, 180, n + 1)[:-1]
        poses = torch.stack([pose_spherical(t, -30., 4.) for t in thetas], 0)
        H = W = args.H or 800
        angle = LEGO_CAMERA_ANGLE_X
    if args.W:
        W = args.W
    focal = .5 * W / np.tan(.5 * angle)
    if args.half_res:
        H, W, focal = H // 2, W // 2, focal / 2.
    return poses, (H, W, focal)


def load_test_set(args):
    """--render_test with a mounted Blender scene (main.py:922-937, 1004-1012): the test split's
    poses, intrinsics and ground-truth RGB (composited on white with --white_bkgd).  Returns
    (poses, (H, W, focal), gt [N,H,W,3] or None); falls back to `load_test_poses` (no GT) when the
    scene's images are not there."""
    from . import blender
    tf = os.path.join(args.datadir, 'transforms_test.json')
    if args.synthetic_poses > 0 or args.dataset_type != 'blender' or not os.path.exists(tf):
        return load_test_poses(args) + (None,)
    with open(tf) as fp:
        first = json.load(fp)['frames'][0]['file_path']
    if not os.path.exists(os.path.join(args.datadir, first + '.png')):
        return load_test_poses(args) + (None,)
    imgs, poses, hwf, _ = blender.load_blender_data(args.datadir, args.half_res, args.testskip, splits=('test',))
    return poses, tuple(hwf), blender.composite(imgs, args.white_bkgd)


# ----------------------------------------------------------------------------------------
# output helpers
# ----------------------------------------------------------------------------------------
def to8b(x):
    return (255 * np.clip(np.asarray(x), 0, 1)).astype(np.uint8)


def write_png(path, rgb8):
    """Minimal 8-bit RGB / RGBA PNG writer (imageio is not a dependency here)."""
    h, w, ch = rgb8.shape
    assert ch in (3, 4) and rgb8.dtype == np.uint8
    rows = np.zeros((h, 1 + w * ch), dtype=np.uint8)      # filter byte 0 in front of every scanline
    rows[:, 1:] = rgb8.reshape(h, w * ch)
    raw = rows.tobytes()

    def chunk(tag, data):
        c = struct.pack('>I', len(data)) + tag + data
        return c + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)

    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 2 if ch == 3 else 6, 0, 0, 0)) +
                chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''))


def mse2psnr(mse):
    return -10. * math.log10(max(float(mse), 1e-30))


# ----------------------------------------------------------------------------------------
# render_path
# ----------------------------------------------------------------------------------------
def teacher_needs_generic(args):
    """True when the teacher the flags describe is not the one the fused kernels are built for (8 x 256, view directions,
    L = 10 / 4, a fine pass: configs/*.txt) or fp32 was asked for: it then renders on generic.GenericNeRF"""
    return (not args.use_viewdirs or args.N_importance <= 0 or args.i_embed != 0 or (args.multires, args.multires_views) != (10, 4)
            or (args.netdepth, args.netwidth, args.netdepth_fine, args.netwidth_fine) != (8, 256, 8, 256) or args.precision == 'fp32')


def build_engine(args, hwf, ckpt, probe_pose=None, probe_rays=None, log=None, probe_poses=None):
    """Engine for the flags of the reference command line.  Every flag that changes what the reference network
    computes is either honoured or refused: a checkpoint trained with another activation / res_scale / depth must
    not render silently wrong images (ResMLP honours them: model/nerf_raybased.py:443-465)."""
    from . import NeRFEngine, PRECISIONS, R2LEngine, R2LError
    H, W, focal = hwf
    auto = args.precision == 'auto'
    # auto: the weights are loaded in fp16x3 -- the mode that packs ANY checkpoint (per-layer scales) -- and choose_precision
    # moves to the generated modes from there; a layer they cannot pack (max|w| outside 2^-12 .. 2^6) then ends in its
    # documented fp16x3 fallback instead of failing r2l_load_weights.  The teacher's `auto` loads in fp16_fp8 and measures from there.
    prec = PRECISIONS.get(args.precision) if not auto else PRECISIONS['fp16x3' if args.model_name in ('R2L', 'nerf_v3.2') else 'fp16_fp8']
    llff_ndc = args.dataset_type == 'llff' and not args.no_ndc
    if args.dataset_type == 'blender':
        near, far = 2., 6.  # main.py:930-931
    elif llff_ndc:
        near, far = 0., 1.  # main.py:917-918
    elif args.trial.near > 0 and args.trial.far > 0:
        near = far = None   # taken from --trial.near / --trial.far below
    else:
        raise R2LError(f'dataset_type={args.dataset_type} with --no_ndc takes near / far from the scene bounds '
                       f'(main.py:913-916), which need the dataset: pass --trial.near and --trial.far')
    if args.trial.near > 0:
        near = args.trial.near
    if args.trial.far > 0:
        far = args.trial.far
    if args.model_name in ('R2L', 'nerf_v3.2'):
        if llff_ndc:
            raise R2LError('the R2L path is built for world-space rays (blender / --no_ndc)')
        if args.plucker or args.learn_depth or args.linear_tail:
            raise R2LError('plucker / learn_depth / linear_tail variants are not built (linear_tail: the reference builds Linear(input_dim, 3) '
                           'on the body output and cannot run it either, model/nerf_raybased.py:532-537)')
        arch = args.trial.body_arch if args.trial.ON else 'mlp'     # model/nerf_raybased.py:499-518: resmlp only under the trial flags
        if arch not in ('resmlp', 'mlp'):
            raise R2LError(f'--trial.body_arch {arch}: resmlp (README.md:51) or mlp')
        # shapes the fused kernels are not built for render on the generic fp32 layer path (generic.py: one launch per nn.Linear):
        # other widths / sample counts / frequencies, --layerwise_netwidths, trial.n_learnable != 2, an odd number of mlp body layers
        generic = (args.netwidth != 256 or args.n_sample_per_ray != 16 or args.multires != 10 or bool(args.layerwise_netwidths)
                   or (args.trial.ON and arch == 'resmlp' and int(args.trial.n_learnable) != 2)
                   or (arch == 'mlp' and ((args.netdepth - 2) % 2 or args.netdepth < 4)) or args.precision == 'fp32')
        if generic:
            if args.precision not in ('auto', 'fp32'):
                raise R2LError(f'--precision {args.precision} is a mode of the fused W256 kernels; this network (netwidth={args.netwidth} '
                               f'n_sample_per_ray={args.n_sample_per_ray} multires={args.multires} layerwise_netwidths={args.layerwise_netwidths!r} '
                               f'n_learnable={args.trial.n_learnable} netdepth={args.netdepth} {arch}) renders on the generic path: --precision auto or fp32')
            from .generic import GenericR2L
            trial = dict(body_arch=arch, n_block=args.trial.n_block, n_learnable=int(args.trial.n_learnable), res_scale=float(args.trial.res_scale),
                         inact=args.trial.inact, outact=args.trial.outact) if args.trial.ON else None
            eng = GenericR2L(H, W, focal, near, far, n_sample=args.n_sample_per_ray, L=args.multires, netdepth=args.netdepth,
                             netwidth=args.netwidth, layerwise_netwidths=args.layerwise_netwidths, act=args.act, use_residual=args.use_residual,
                             trial=trial)
            eng.load_state_dict(ckpt['network_fn_state_dict'])
            if log:
                log(f'[precision] {args.precision}: this network is outside the fused kernels\' shapes -> generic fp32 layer path '
                    f'({len(eng.plan)} Linear launches per chunk, {eng.flops_per_ray / 1e6:.2f} MFLOP/ray)')
            return 'R2L', eng
        if args.precision == 'fp16_mix':
            raise R2LError('--precision fp16_mix is a mode of the NeRF teacher\'s fine network (the R2L student has fp16_split / fp16_split8)')
        acts = (args.act.lower(), args.trial.inact.lower(), args.trial.outact.lower())
        for a in acts:
            if a not in R2LEngine.ACT_SLOPES:
                raise R2LError(f'activation {a!r}: the reference knows relu, lrelu and none (model/nerf_raybased.py:468-476)')
        if arch == 'mlp' and acts[0] == 'none':
            raise R2LError('--trial.body_arch mlp with --act none: the reference cannot build that network either (nn.Sequential of None)')
        if acts != ('relu', 'relu', 'none') or arch == 'mlp':
            # lrelu / none variants (model/nerf_raybased.py:468-476) and the plain-MLP body render in the compiler-scheduled fp16x3 only
            if args.precision not in ('auto', 'fp16x3', 'fp16x1'):
                raise R2LError(f'act={acts[0]} trial.inact={acts[1]} trial.outact={acts[2]} body_arch={arch}: --precision {args.precision} is a '
                               f'generated kernel built for relu / relu / none ResMLP blocks; use --precision auto (or fp16x3)')
        # trial.n_block is read in the resmlp branch only (model/nerf_raybased.py:503-514); the mlp body has netdepth - 2 layers (:515-518)
        n_block = args.trial.n_block if (arch == 'resmlp' and args.trial.n_block > 0) else (args.netdepth - 2) // 2
        eng = R2LEngine(H, W, focal, near, far, n_sample=args.n_sample_per_ray, L=args.multires,
                        width=args.netwidth, n_block=n_block, use_residual=args.use_residual, precision=prec,
                        res_scale=float(args.trial.res_scale), act=acts[0], inact=acts[1], outact=acts[2], body_arch=arch)
        eng.load_state_dict(ckpt['network_fn_state_dict'])
        if args.precision in ('fp16_split', 'fp16_split8') and getattr(args, 'split_block', -1) >= 0:
            eng.set_split_block(args.split_block)
            if log:
                log(f'[precision] {args.precision}: blocks [0, {eng.split_block}) in three fp16 passes, blocks [{eng.split_block}, {eng.n_block}) with '
                    f'{"e4m3" if args.precision.endswith("8") else "bf6"} terms (--split_block)')
        if auto:
            if probe_pose is None and probe_rays is None:
                raise R2LError('--precision auto needs a pose or rays to measure the activation ranges with')
            # ranges from the first pose; the verification of the rung and the split measurement on up to three poses spanning the path
            plist = [p for p in (probe_poses or []) if p is not None]
            name, top = eng.choose_precision(c2w=plist if (probe_rays is None and len(plist) > 1) else probe_pose, rays=probe_rays)
            from . import dist as D
            D.agree_precision(eng)        # several ranks: rank 0's rung and split on every rank (each measured its own; they must not differ)
            name = eng.precision_name
            if log and top is None:
                log(f'[precision] auto: {eng.auto_note} -> {name}')
            elif log:
                log(f'[precision] auto: activations of every ray of the first frame up to {eng.stream_max:.2f} (exponent {top}; '
                    f'fp16_fp8 up to {eng.AUTO_MAX_ABS:g}, fp16_e4m3 up to {eng.AUTO_MAX_ABS_E4M3:g}) -> {name}')
                if getattr(eng, 'auto_verify', None) is not None:
                    log(f'[precision] auto: the rung those limits name is {eng.auto_verify:.1e} from three passes on every ray of {max(1, len(plist))} probe frame(s) '
                        f'(limit {eng.AUTO_VERIFY_MAX_DIFF:g})' + ('' if eng.auto_verify <= eng.AUTO_VERIFY_MAX_DIFF else ': measured rungs instead'))
                if getattr(eng, 'auto_split', None):
                    tried = '; '.join(f"{'e4m3' if m.endswith('8') else 'bf6'} terms behind them: " + ', '.join(f'{k}: {v:.1e}' for k, v in t.items())
                                      for m, t in eng.auto_split.items())
                    log(f'[precision] auto: leading blocks in three passes, largest rgb difference from three passes everywhere on every ray of '
                        f'{max(1, len(plist))} probe frame(s) -- {tried} (limit {eng.AUTO_SPLIT_MAX_DIFF:g}) -> ' +
                        (f'{name} at block {eng.split_block} of {eng.n_block}' if name.startswith('fp16_split') else name))
        return 'R2L', eng
    if args.model_name == 'nerf':
        # the fused teacher kernels are the 8 x 256 NeRF with view directions, L = 10 / 4 and a fine pass (configs/*.txt); every other
        # network create_nerf builds (main.py:407-453) renders on the generic fp32 layer path (generic.GenericNeRF)
        if teacher_needs_generic(args):
            if args.precision not in ('auto', 'fp32'):
                raise R2LError(f'--precision {args.precision} is a mode of the fused 8 x 256 teacher kernels; netdepth/netwidth(_fine) = '
                               f'{args.netdepth}/{args.netwidth}/{args.netdepth_fine}/{args.netwidth_fine}, i_embed={args.i_embed}, multires='
                               f'{args.multires}/{args.multires_views}, use_viewdirs={args.use_viewdirs}, N_importance={args.N_importance} '
                               f'renders on the generic path: --precision auto or fp32')
            if args.i_embed not in (0, -1):
                raise R2LError(f'i_embed={args.i_embed}: 0 (positional encoding) or -1 (none), utils/run_nerf_raybased_helpers.py:59-74')
            if args.N_importance > 0 and 'network_fine_state_dict' not in ckpt:
                raise KeyError("checkpoint lacks 'network_fine_state_dict'")
            from .generic import GenericNeRF
            eng = GenericNeRF(H, W, focal, near, far, N_samples=args.N_samples, N_importance=args.N_importance, multires=args.multires,
                              multires_views=args.multires_views, i_embed=args.i_embed, netdepth=args.netdepth, netwidth=args.netwidth,
                              netdepth_fine=args.netdepth_fine, netwidth_fine=args.netwidth_fine, use_viewdirs=args.use_viewdirs,
                              white_bkgd=args.white_bkgd, lindisp=args.lindisp, ndc=llff_ndc)
            eng.load_state_dicts(ckpt['network_fn_state_dict'], ckpt.get('network_fine_state_dict'))
            if log:
                log(f'[precision] {args.precision}: this teacher is outside the fused kernels\' shapes -> generic fp32 layer path '
                    f'({eng.flops_per_ray / 1e6:.1f} MFLOP/ray)')
            return 'nerf', eng
        if 'network_fine_state_dict' not in ckpt:
            raise KeyError("checkpoint lacks 'network_fine_state_dict'")
        if args.precision in ('fp16_e4m3', 'fp16_split', 'fp16_split8'):
            raise R2LError(f'--precision {args.precision} is a mode of the R2L student (the teacher has fp16x3, fp16x3_asm, fp16_fp8, fp16x1)')
        eng = NeRFEngine(H, W, focal, near, far, N_samples=args.N_samples, N_importance=args.N_importance,
                         multires=args.multires, multires_views=args.multires_views, white_bkgd=args.white_bkgd,
                         precision=prec, ndc=llff_ndc, lindisp=args.lindisp)  # main.py:160-162, 525-528, 679-680
        eng.load_state_dicts(ckpt['network_fn_state_dict'], ckpt['network_fine_state_dict'])
        if auto:
            # one fp16 pass / the chain's bf6 terms under fixed activation exponents: measured against fp16x3 on rays of the job's
            # own poses (first, middle, last of the path when the caller names them), and watched afterwards (render_path)
            from .teacher import get_rays
            if probe_rays is not None:
                sets = [tuple(t.to(eng.device, torch.float32) for t in probe_rays)]
            else:
                plist = list(probe_poses) if probe_poses is not None else [probe_pose]
                sets = [tuple(t.reshape(-1, 3) for t in get_rays(H, W, focal, torch.as_tensor(p)[:3, :4], device=eng.device)) for p in plist]
            name, diff = eng.choose_precision(sets)
            if log:
                lims = {'fp16x1': eng.AUTO_MAX_DIFF_X1, 'fp16_fp8': eng.AUTO_MAX_DIFF, 'fp16_mix': eng.AUTO_MAX_DIFF_MIX, 'fp16x3_asm': eng.AUTO_MAX_DIFF_X3ASM}
                tried = ', '.join(f'{k} {v:.1e} (limit {lims.get(k, 0):.0e})' for k, v in eng.auto_diffs.items())
                log(f'[precision] auto: largest rgb / acc difference from fp16x3 on {min(4096, sets[0][0].shape[0])} rays of each of {len(sets)} probe '
                    f'frame(s) (fp16_mix = coarse fp16x3_asm + fine bf6 chain with two three-pass layers: from fp16x3_asm for both, up to '
                    f'{eng.MIX_PROBE_RAYS} rays; fp16x3_asm: stage by stage): {tried} -> {name}')
        return 'nerf', eng
    raise R2LError(f'model_name={args.model_name} is not a render path of this build')


class _ImageWriter:
    """PNG encoding off the render loop (the reference writes with imageio inside its loop, main.py:337-341; here rank 0
    would hold the next collective while it deflates 1.9 MB in Python).  Frames are handed over as pinned host tensors whose
    copy from the device is still in flight on a side stream, with the event behind that copy; a worker thread waits for the
    event and encodes (zlib releases the GIL) while the loop goes on rendering.  Errors surface in close()."""

    def __init__(self, n_threads=4):
        import queue
        import threading
        self.q = queue.Queue()
        self.err = []
        self.threads = [threading.Thread(target=self._work, daemon=True) for _ in range(n_threads)]
        for t in self.threads:
            t.start()

    def _work(self):
        while True:
            item = self.q.get()
            if item is None:
                return
            path, img, ready = item
            try:
                if ready is not None:
                    ready.synchronize()          # the event behind the D2H copy of this frame
                write_png(path, to8b(img.numpy() if torch.is_tensor(img) else img))
            except Exception as e:               # surfaced by close()
                self.err.append(e)

    def put(self, path, img, ready=None):
        self.q.put((path, img, ready))

    def close(self):
        for _ in self.threads:
            self.q.put(None)
        for t in self.threads:
            t.join()
        if self.err:
            raise self.err[0]


def render_path(render_poses, hwf, kind, eng, gt_imgs=None, savedir=None, log=print, given_rays=None, frames_per_batch=None,
                stats=None, watch_every=8):
    """main.py:189-398 for the R2L and nerf branches: render, per-frame timing lines, PSNR / SSIM when GT is given.

    The loop is the one bench.py times ("batch >= 8 frames per collective"): frames go in batches of
    `frames_per_batch` (default: the world size, i.e. one frame at N = 1) -- every rank renders its row shard of ALL frames of
    the batch in one launch (`render_batch`: world x H/world rows = one frame's worth of ray tiles per rank, the grid bench.py
    fills), ONE `r2l_gather_image` assembles them straight into the frame stack, the range check (one all-reduce) runs once
    per batch, and the host synchronises once per batch for the reference's timing line.  PNGs are encoded on worker threads
    from pinned host copies made on a side stream; PSNR / SSIM stay on the device until the loop is over.
    `given_rays` = (all_rays_o, all_rays_d) [N, H*W, 3] replaces the camera rays (--given_render_path_rays,
    main.py:207-230: the DONERF test path through PointSampler.sample_train).
    `stats` (dict, optional) receives the loop's own numbers: render_loop_s (first launch to last sync, image encoding
    excluded), batches, collectives, rerenders.
    Teacher frames in a fast mode (fp16x1 / fp16_fp8, chosen once by `--precision auto` or by flag) are watched: every
    `watch_every`-th frame (and the first) a rank re-renders NeRFEngine.WATCH_RAYS of its rays in fp16x3 and compares rgb / acc /
    depth (NeRFEngine.spot_check); a miss on any rank moves every rank one rung down the ladder and the frame is rendered
    again (`stats['watch']`).  0 switches the watch off."""
    from . import dist as D
    import torch.distributed as tdist
    H, W, focal = hwf
    world = tdist.get_world_size() if tdist.is_initialized() else 1
    rank = tdist.get_rank() if tdist.is_initialized() else 0
    r0, r1 = D.row_shard(H, rank, world)
    n_local = (r1 - r0) * W
    from .metrics import ssim_hwc
    n_frames = len(given_rays[0]) if given_rays is not None else len(render_poses)
    B = max(1, int(frames_per_batch or world))
    # the frame stack is allocated once and every batch is gathered straight into its slots: the collective's own buffer is
    # reused from call to call (dist.RowGather.gather), so keeping views of it would keep N copies of the LAST frame
    rgbs = torch.empty((n_frames, H, W, 3), dtype=torch.float32, device=eng.device)
    local = torch.empty((B, n_local, 3), dtype=torch.float32, device=eng.device) if world > 1 else None
    pose_dev = None
    if given_rays is None and kind == 'R2L' and n_frames:
        pose_dev = torch.stack([torch.as_tensor(p)[:3, :4].float() for p in render_poses], 0).contiguous().to(eng.device)

    def render_local(i0, nb):
        """row shard of frames i0 .. i0 + nb - 1 into local[:nb] (one rank: straight into the frame stack, nothing to assemble)"""
        dst = local[:nb] if world > 1 else rgbs[i0:i0 + nb].view(nb, n_local, 3)
        if pose_dev is not None:
            eng.render_batch(pose_dev[i0:i0 + nb], rows=(r0, r1), out=dst)
            return
        for f in range(nb):
            i = i0 + f
            ro = rd = None
            if given_rays is not None:
                ro = given_rays[0][i].reshape(H, W, 3)[r0:r1].reshape(-1, 3).contiguous().to(eng.device, torch.float32)
                rd = given_rays[1][i].reshape(H, W, 3)[r0:r1].reshape(-1, 3).contiguous().to(eng.device, torch.float32)
                got = eng.render_rays(ro, rd)
            else:
                got = eng.render(render_poses[i][:3, :4], rows=(r0, r1))
            if watching and i % watch_every == 0:
                got = watched(i, ro, rd, got)
            dst[f].copy_(got if kind == 'R2L' else got['rgb_map'])

    # every rung `auto` can choose for the R2L student under an rgb watch (R2LEngine.spot_check_rgb): the split rungs -- `auto` measured on
    # the probe frames how many leading blocks need three passes -- every watch_every-th batch (a sample of the rank's own rays of the
    # batch's first frame with the split and with three passes everywhere; a miss on any rank moves half of the low-precision part to
    # three passes on every rank); the whole-network rungs fp16_fp8 / fp16_e4m3 (round 6) every WHOLE_WATCH_EVERY-th batch against a second
    # context holding the weights in three passes (a miss sends the network to the measured split rungs: rank 0's choice on every
    # rank, then checked on every rank's rays by the same loop).  The batch is rendered again after every change.
    split_watching = kind == 'R2L' and watch_every > 0 and hasattr(eng, 'spot_check_rgb')
    split_watch = {'checks': 0, 'fallbacks': [], 'worst': 0.0}
    whole_every = max(1, int(getattr(eng, 'WHOLE_WATCH_EVERY', 16) * watch_every / 8)) if watch_every > 0 else 0

    def watch_due(n_batches):
        mode = eng.watched_mode() if split_watching else None
        return (mode == 'split' and n_batches % watch_every == 0) or (mode == 'whole' and n_batches % whole_every == 0)

    def watch_split(i0, nb):
        from .teacher import get_rays
        again = 0
        for _ in range(6):
            # every rank enters the collective whatever its local mode is (a rank that had left the rung would otherwise
            # miss the all-reduce its peers are in); a rank with nothing to watch reports "good"
            mode = eng.watched_mode()
            ok, d = True, 0.0
            if mode is not None:
                if given_rays is not None:
                    ro = given_rays[0][i0].reshape(H, W, 3)[r0:r1].reshape(-1, 3).contiguous().to(eng.device, torch.float32)
                    rd = given_rays[1][i0].reshape(H, W, 3)[r0:r1].reshape(-1, 3).contiguous().to(eng.device, torch.float32)
                else:
                    ro, rd = (t.reshape(-1, 3) for t in get_rays(H, W, focal, torch.as_tensor(render_poses[i0])[:3, :4], rows=(r0, r1), device=eng.device))
                ok, d = eng.spot_check_rgb(ro, rd)
                split_watch['checks'] += 1
                split_watch['worst'] = max(split_watch['worst'], d)
            bad = 0 if ok else 1
            if world > 1:
                t = torch.tensor([bad], dtype=torch.int32, device=eng.device if tdist.get_backend() == 'nccl' else 'cpu')
                tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
                bad = int(t.item())
            if not bad:
                break
            was_mode, was = eng.precision_name, eng.split_block
            if mode == 'whole':
                now = eng.step_down_whole(ro, rd)
                D.agree_precision(eng)                 # rank 0's measured rung on every rank; the loop checks it on every rank's rays next
                now = eng.precision_name
                limit, n_w = eng.WHOLE_WATCH_MAX_DIFF, eng.WHOLE_WATCH_RAYS
            elif mode == 'split':
                now = eng.step_down_split()
                limit, n_w = eng.SPLIT_WATCH_MAX_DIFF, eng.SPLIT_WATCH_RAYS
            else:                                      # a peer missed: follow its rung (it steps down deterministically from the shared state)
                now = eng.precision_name
                limit, n_w = eng.SPLIT_WATCH_MAX_DIFF, eng.SPLIT_WATCH_RAYS
            split_watch['fallbacks'].append({'frame': i0, 'from': was_mode, 'split': was, 'to': eng.split_block if now.startswith('fp16_split') else now, 'diff': d})
            if rank == 0:
                what = f'with low-precision terms from block {was} on' if mode == 'split' else f'{was_mode} is'
                log(f'[precision] frame {i0}: {what} {d:.1e} from three passes on {n_w} of its rays (limit {limit:g}) -> ' +
                    (f'{now} at block {eng.split_block}' if now.startswith('fp16_split') else now) + '; batch rendered again')
            eng.render_checked(lambda: render_local(i0, nb), check=check)
            again += 1
        return again

    # the teacher's fast modes under watch: fp16x1 / fp16_fp8 were chosen on a probe; every watch_every-th
    # frame is checked against fp16x3 on a sample of its own rays, agreed between the ranks, with fallback + re-render
    watching = kind != 'R2L' and watch_every > 0 and hasattr(eng, 'spot_check')
    watch = {'checks': 0, 'fallbacks': [], 'worst': {}}
    if kind != 'R2L' and hasattr(eng, 'set_skip_rgb0'):
        # this loop keeps rgb only (main.py:277-282 drops render()'s extras): the coarse pass runs without its view branch when its mode has
        # that build (fp16x3_asm, the coarse mode of every trained teacher) -- bit for bit the same frames, 17 % fewer coarse MACs
        eng.set_skip_rgb0(True)
        if rank == 0 and eng._rgb0_skipped():
            log('[precision] rgb only: the coarse pass runs without its view branch and fine-pass tiles without a positive density skip theirs '
                '(rgb0 is not produced; rgb / disp / acc are bit for bit the same)')

    def watched(i, ro, rd, got):
        from .teacher import get_rays
        for _ in range(3):
            if eng.precision_name == 'fp16x3':          # the mode the watch compares with (fp16x3_asm is watched too since round 6)
                break
            if ro is None:
                ro, rd = (t.reshape(-1, 3) for t in get_rays(H, W, focal, torch.as_tensor(render_poses[i])[:3, :4], rows=(r0, r1),
                                                             device=eng.device))
            ok, d = eng.spot_check(ro, rd, got)
            watch['checks'] += 1
            for k, v in d.items():
                watch['worst'][k] = max(watch['worst'].get(k, 0.), v)
            bad = 0 if ok else 1
            if world > 1:     # every rank checks the same frames: one 4-byte all-reduce per check, all ranks act alike
                t = torch.tensor([bad], dtype=torch.int32, device=eng.device if tdist.get_backend() == 'nccl' else 'cpu')
                tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
                bad = int(t.item())
            if not bad:
                break
            was = eng.precision_name
            now = eng.step_down()
            watch['fallbacks'].append({'frame': i, 'from': was, 'to': now, 'diffs': d})
            if rank == 0:
                log(f'[precision] frame {i}: {was} is {d} from fp16x3 on {eng.WATCH_RAYS} of its rays (limits '
                    f'{eng.AUTO_MAX_DIFF_X1 if was == "fp16x1" else eng.AUTO_MAX_DIFF:.0e} x (1, 1, far)) -> {now}; frame rendered again')
            got = eng.render_rays(ro, rd) if given_rays is not None else eng.render(render_poses[i][:3, :4], rows=(r0, r1))
        return got

    if kind == 'R2L' and world > 1 and n_frames > 0:
        # an explicit fp16_fp8 measures its activation ranges on the first render's own rays: one agreed set for all row
        # shards (`auto` has measured every ray of a whole frame on every rank already, the all-reduce then only confirms it)
        render_local(0, 1)
        D.agree_act_exponents(eng)
    writer = _ImageWriter() if (savedir is not None and rank == 0) else None
    copy_stream = torch.cuda.Stream(device=eng.device) if writer is not None else None
    # the host copy of the frame stack, pinned, allocated once (a pinned allocation per batch cost 0.3 ms of every 10 ms frame):
    # every batch is copied into its slots on the side stream; the encoder threads and the caller (`stats['host_frames']`) read it
    host_stack = torch.empty((n_frames, H, W, 3), dtype=torch.float32, pin_memory=True) if writer is not None else None
    mse_dev, ssim_dev = [], []
    n_coll = n_again = n_batches = 0
    check = (lambda: D.check_ranges(eng, log=log if rank == 0 else None)) if kind == 'R2L' else (lambda: None)
    torch.cuda.synchronize()
    t_loop = time.time()
    for i0 in range(0, n_frames, B):
        nb = min(B, n_frames - i0)
        t0 = time.time()
        if kind == 'R2L':
            # did this batch's rays stay inside the range the low-precision scales were measured for?  If not: measured,
            # raised, or the context falls back (`auto`), and the batch is rendered again -- one check per batch
            _, again = eng.render_checked(lambda: render_local(i0, nb), check=check)
            n_again += again
            if watch_due(n_batches):
                n_again += watch_split(i0, nb)
        else:
            render_local(i0, nb)
        if world > 1:
            D.gather_rows(local[:nb], H, W, world, out=rgbs[i0:i0 + nb])
            n_coll += 1
        n_batches += 1
        if writer is not None:      # D2H on a side stream behind the gather; the worker waits for the event, the loop does not
            done = torch.cuda.Event()
            copy_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(copy_stream):
                host_stack[i0:i0 + nb].copy_(rgbs[i0:i0 + nb], non_blocking=True)
                done.record(copy_stream)
            for f in range(nb):
                writer.put(os.path.join(savedir, f'{i0 + f:03d}.png'), host_stack[i0 + f], done)
        if gt_imgs is not None:
            for f in range(nb):
                gt = gt_imgs[i0 + f].to(rgbs.device)
                mse_dev.append(torch.mean((rgbs[i0 + f] - gt) ** 2))
                ssim_dev.append(ssim_hwc(rgbs[i0 + f], gt))   # main.py:334-335
                if writer is not None:  # main.py:340-341
                    writer.put(os.path.join(savedir, f'{i0 + f:03d}_gt.png'), gt_imgs[i0 + f].cpu().numpy())
        torch.cuda.synchronize()       # the reference's per-frame bracket (main.py:273-310), once per batch here
        if rank == 0:
            dt = (time.time() - t0) / nb
            for f in range(nb):
                log(f'[#{i0 + f}] frame, rendering done, time for this frame: {dt:.4f}s')
    t_loop = time.time() - t_loop
    if writer is not None:
        writer.close()
    if stats is not None:
        stats.update(render_loop_s=t_loop, frames=n_frames, frames_per_batch=B, batches=n_batches, collectives=n_coll,
                     rerenders=n_again, world=world, rows_per_rank=r1 - r0)
        if watching:
            stats['watch'] = dict(watch, every=watch_every, precision=eng.precision_name)
        if split_watching and split_watch['checks']:
            stats['split_watch'] = dict(split_watch, every=watch_every, whole_every=whole_every, precision=eng.precision_name, split_block=eng.split_block)
        if host_stack is not None:
            stats['host_frames'] = host_stack          # complete: writer.close() has waited for every copy
    misc = {}
    if gt_imgs is not None:
        misc['test_psnr'] = mse2psnr(torch.mean((rgbs - gt_imgs.to(rgbs.device)) ** 2))
        misc['test_psnr_v2'] = float(np.mean([mse2psnr(m) for m in mse_dev]))
        misc['test_ssim'] = float(np.mean([float(v) for v in ssim_dev]))
    return rgbs, misc


def main(argv=None):
    from . import dist as D
    args = parse_args(argv)
    if not args.render_only:
        raise SystemExit('this front-end implements the --render_only path (training is out of scope)')
    if not args.pretrained_ckpt:
        raise SystemExit('--pretrained_ckpt is required with --render_only')
    rank, local_rank, world = D.init()
    torch.cuda.set_device(D.local_device(local_rank))
    log = print if rank == 0 else (lambda *a, **k: None)
    ckpt = load_checkpoint(args.pretrained_ckpt)
    log(f'Load pretrained ckpt successfully: "{args.pretrained_ckpt}".')
    gt = None
    if args.render_test:
        poses, hwf, gt = load_test_set(args)
    else:
        poses, hwf = load_test_poses(args)
    given = None
    if args.given_render_path_rays:  # main.py:207-213
        loaded = torch.load(args.given_render_path_rays, map_location='cpu')
        given = (loaded['all_rays_o'].float(), loaded['all_rays_d'].float())
        if 'gt_imgs' in loaded:
            gt = loaded['gt_imgs'].float()
        log(f'Use given render_path rays: "{args.given_render_path_rays}"')
    if args.render_factor != 0:  # main.py:197-201: render downsampled, compare with the top-left crop
        H_, W_, f_ = hwf
        hwf = (int(H_ / args.render_factor), int(W_ / args.render_factor), f_ / args.render_factor)
        if gt is not None:
            gt = gt[:, :hwf[0], :hwf[1]]
    kind, eng = build_engine(args, hwf, ckpt, probe_pose=None if given is not None else poses[0][:3, :4],
                             probe_rays=None if given is None else (given[0][0].reshape(-1, 3), given[1][0].reshape(-1, 3)),
                             probe_poses=None if given is not None else [poses[k][:3, :4] for k in sorted({0, len(poses) // 2, len(poses) - 1})],
                             log=log)
    outdir = args.outdir or os.path.join(args.basedir, args.expname or 'render', 'gen_img')
    if rank == 0:
        os.makedirs(outdir, exist_ok=True)
    if args.benchmark:  # main.py:1124-1133: timeit(100) of render_func(model, pose) on the first pose
        if kind != 'R2L':
            raise SystemExit('--benchmark times render_func, the R2L path (main.py:401-404)')
        H, W, _ = hwf
        pose = poses[0][:3, :4]
        for _ in range(3):
            eng.render(pose)
        torch.cuda.synchronize()
        t_ = time.time()
        for _ in range(100):
            eng.render(pose)
        torch.cuda.synchronize()
        dt = (time.time() - t_) / 100
        log(f'render_func(model, pose): {dt * 1e3:.3f} ms per {H}x{W} frame over 100 runs ({H * W / dt:.3e} rays/s)')
        return 0
    log('RENDER ONLY')
    t_ = time.time()
    st = {}
    with torch.no_grad():
        rgbs, misc = render_path(poses, hwf, kind, eng, gt_imgs=gt, savedir=outdir, log=log, given_rays=given, stats=st,
                                 frames_per_batch=args.frames_per_batch or None, watch_every=args.watch_every)
    dt = time.time() - t_
    if rank == 0:
        np.save(os.path.join(outdir, 'rgbs.npy'), st['host_frames'].numpy() if 'host_frames' in st else rgbs.cpu().numpy())
        H, W, _ = hwf
        # the loop bench.py times (render + range check + gather, synchronised per batch); image encoding runs beside it
        log(f'Render loop: {len(rgbs)} view(s) {H}x{W} on {world} GPU(s) in {st["render_loop_s"]:.3f}s = '
            f'{len(rgbs) * H * W / max(st["render_loop_s"], 1e-9):.3e} rays/s ({st["batches"]} batch(es) of {st["frames_per_batch"]} '
            f'frame(s), {st["collectives"]} collective(s), {st["rerenders"]} re-render(s))')
        log(f'Rendered {len(rgbs)} view(s) {H}x{W} on {world} GPU(s) in {dt:.2f}s '
            f'({len(rgbs) * H * W / dt:.3e} rays/s incl. host I/O)')
        if st.get('split_watch'):
            w_ = st['split_watch']
            every = w_['every'] if str(w_['precision']).startswith('fp16_split') else w_.get('whole_every', w_['every'])
            log(f"[precision] rgb watch: {w_['checks']} spot check(s) against three passes (every {every} batches), worst {w_['worst']:.2e}, "
                f"{len(w_['fallbacks'])} fallback(s); at the end: {w_['precision']}" + (f" at block {w_['split_block']}" if w_['precision'].startswith('fp16_split') else ''))
        if st.get('watch', {}).get('checks'):
            w_ = st['watch']
            log(f"[precision] watch: {w_['checks']} spot check(s) against fp16x3 (every {w_['every']} frames, {eng.WATCH_RAYS} rays), worst "
                f"{ {k: float('%.2e' % v) for k, v in w_['worst'].items()} }, {len(w_['fallbacks'])} fallback(s); mode at the end: {w_['precision']}")
        if 'test_psnr' in misc:
            log(f"[TEST] TestPSNR {misc['test_psnr']:.4f} TestPSNRv2 {misc['test_psnr_v2']:.4f} "
                f"TestSSIM {misc['test_ssim']:.4f}")
        log(f'Save renders: "{outdir}"')
    return 0
