"""GPU: the reference's command line end to end on synthetic checkpoints (R2L and teacher),
renders compared with the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_main(args, cwd=ROOT):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py')] + args, cwd=cwd, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_r2l_render_only_cli(pkg, tmp_path):
    from efficient_nerf_amd import frontend as fe
    sd = O.make_r2l_state(seed=4, netdepth=6)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    out = str(tmp_path / 'out')
    log = run_main(['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16',
                    '--netwidth', '256', '--netdepth', '6', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp',
                    '--pretrained_ckpt', ck, '--render_only', '--render_test', '--testskip', '1', '--screen',
                    '--synthetic_poses', '2', '--H', '64', '--outdir', out])
    assert 'RENDER ONLY' in log and 'Load pretrained ckpt successfully' in log
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    H = 32  # half_res of --H 64
    assert rgbs.shape == (2, H, H, 3) and os.path.exists(os.path.join(out, '001.png'))
    focal = O.focal_from_angle(64) / 2.
    for i, c2w in enumerate(O.novel_poses(2)):
        ref = O.r2l_render(sd, H, H, focal, c2w).view(H, H, 3).numpy()
        assert np.abs(rgbs[i] - ref).max() <= 1e-4


def test_r2l_render_test_against_mounted_scene(pkg, tmp_path):
    """--render_test with a Blender-layout scene on disk (transforms_test.json + RGBA PNGs): the
    test split's own poses are rendered and PSNR / SSIM against the frames are reported
    (main.py:331-335, 1080).  The frames are the CPU oracle's renders quantised to 8 bit, so the
    report must come out at the quantisation floor of to8b (truncation: MSE = (1/255)^2/3 -> 52.9 dB)."""
    import json
    import re
    from efficient_nerf_amd import frontend as fe
    H = 32
    sd = O.make_r2l_state(seed=6, netdepth=4)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    scene = tmp_path / 'scene'
    os.makedirs(scene / 'test')
    focal = O.focal_from_angle(H)
    poses = O.novel_poses(6)
    frames = []
    for i, c2w in enumerate(poses):
        rgb = O.r2l_render(sd, H, H, focal, c2w).view(H, H, 3).numpy()
        rgba = np.concatenate([fe.to8b(rgb), np.full((H, H, 1), 255, np.uint8)], -1)
        fe.write_png(str(scene / 'test' / f'r_{i}.png'), rgba)
        frames.append({'file_path': f'./test/r_{i}', 'transform_matrix': c2w.tolist()})
    with open(scene / 'transforms_test.json', 'w') as fp:
        json.dump({'camera_angle_x': O.LEGO_CAMERA_ANGLE_X, 'frames': frames}, fp)
    out = str(tmp_path / 'out')
    log = run_main(['--model_name', 'R2L', '--config', 'configs/lego_noview_800x800.txt', '--datadir', str(scene),
                    '--n_sample_per_ray', '16', '--netwidth', '256', '--netdepth', '4', '--use_residual', '--trial.ON',
                    '--trial.body_arch', 'resmlp', '--pretrained_ckpt', ck, '--render_only', '--render_test',
                    '--testskip', '2', '--outdir', out])
    m = re.search(r'TestPSNR ([0-9.]+) TestPSNRv2 ([0-9.]+) TestSSIM ([0-9.]+)', log)
    assert m, log[-1500:]
    assert abs(float(m.group(1)) - 52.9) < 0.3 and abs(float(m.group(2)) - 52.9) < 0.3 and float(m.group(3)) > 0.995, m.group(0)
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    assert rgbs.shape == (3, H, H, 3) and os.path.exists(os.path.join(out, '002_gt.png'))  # frames[::2]


def test_given_render_path_rays_and_render_factor(pkg, tmp_path):
    """--given_render_path_rays (main.py:207-230: rays from a .pt file through the sample_train
    path) fed with the camera rays of two poses must reproduce the pose render; --render_factor 2
    halves H, W and focal (main.py:197-201)."""
    from efficient_nerf_amd import frontend as fe
    sd = O.make_r2l_state(seed=9, netdepth=4)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    H = 32  # --H 64 with the config's half_res
    focal = O.focal_from_angle(64) / 2.
    poses = O.novel_poses(2)
    ro, rd = zip(*[O.rays_from_dirs(O.camera_dirs(H, H, focal), c[:3, :4]) for c in poses])
    rays = str(tmp_path / 'rays.pt')
    torch.save({'all_rays_o': torch.stack([r.reshape(-1, 3) for r in ro]), 'all_rays_d': torch.stack([r.reshape(-1, 3) for r in rd])}, rays)
    base = ['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
            '--netdepth', '4', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--pretrained_ckpt', ck,
            '--render_only', '--synthetic_poses', '2', '--H', '64']
    run_main(base + ['--outdir', str(tmp_path / 'a')])
    run_main(base + ['--outdir', str(tmp_path / 'b'), '--given_render_path_rays', rays])
    a, b = np.load(tmp_path / 'a' / 'rgbs.npy'), np.load(tmp_path / 'b' / 'rgbs.npy')
    assert a.shape == b.shape == (2, H, H, 3) and np.abs(a - b).max() <= 2e-6
    log = run_main(base + ['--benchmark'])  # main.py:1124-1133
    assert 'render_func(model, pose)' in log and 'rays/s' in log
    run_main(base + ['--outdir', str(tmp_path / 'c'), '--render_factor', '2'])
    c = np.load(tmp_path / 'c' / 'rgbs.npy')
    assert c.shape == (2, H // 2, H // 2, 3)
    ref = O.r2l_render(sd, H // 2, H // 2, focal / 2., poses[0]).view(H // 2, H // 2, 3).numpy()
    assert np.abs(c[0] - ref).max() <= 1e-4


def test_teacher_render_only_cli(pkg, tmp_path):
    from efficient_nerf_amd import frontend as fe
    t0, t1 = O.make_teacher_state(1), O.make_teacher_state(2)
    ck = str(tmp_path / 'nerf.tar')
    fe.save_checkpoint(ck, t0, t1)
    out = str(tmp_path / 'out')
    log = run_main(['--model_name', 'nerf', '--config', 'configs/lego.txt', '--pretrained_ckpt', ck, '--render_only',
                    '--render_test', '--testskip', '1', '--synthetic_poses', '1', '--H', '16', '--outdir', out])
    # the CLI default is --precision auto: for the teacher the candidates are measured against fp16x3 first, fastest first
    assert '[precision] auto: largest rgb / acc difference from fp16x3 on' in log and '[precision] watch: 1 spot check(s)' in log and log.split('[precision] auto')[1].splitlines()[0].endswith('-> fp16x1'), log
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    H = 8
    assert rgbs.shape == (1, H, H, 3)
    ref = O.teacher_render(t0, t1, H, H, O.focal_from_angle(16) / 2., O.novel_poses(1)[0], white_bkgd=True)
    assert np.abs(rgbs[0].reshape(-1, 3) - ref['rgb_map'].numpy()).max() <= 1e-4


def test_teacher_llff_ndc_cli(pkg, tmp_path):
    """--dataset_type llff without --no_ndc: render(..., ndc=True) with near, far = 0, 1 (main.py:160-162, 525-528,
    917-918), here also with --lindisp (main.py:679-680); poses / intrinsics synthetic (the LLFF loader is not built)."""
    from efficient_nerf_amd import frontend as fe
    t0, t1 = O.make_teacher_state(1), O.make_teacher_state(2)
    ck = str(tmp_path / 'nerf.tar')
    fe.save_checkpoint(ck, t0, t1)
    focal = .5 * 14 / np.tan(.5 * 0.6911112070083618)
    common = ['--model_name', 'nerf', '--use_viewdirs', '--N_importance', '128', '--pretrained_ckpt', ck, '--render_only',
              '--render_test', '--synthetic_poses', '1', '--H', '10', '--W', '14']
    out = str(tmp_path / 'ndc')
    run_main(common + ['--dataset_type', 'llff', '--outdir', out])
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    assert rgbs.shape == (1, 10, 14, 3)
    ref = O.teacher_render(t0, t1, 10, 14, focal, O.novel_poses(1)[0], ndc=True, near=0., far=1., white_bkgd=False)
    assert np.abs(rgbs[0].reshape(-1, 3) - ref['rgb_map'].numpy()).max() <= 1e-4
    # --lindisp reaches the engine (with the blender bounds: near = 0 of the NDC path makes 1/near infinite in the reference too)
    out = str(tmp_path / 'lindisp')
    run_main(common + ['--dataset_type', 'blender', '--lindisp', '--outdir', out])
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    ref = O.teacher_render(t0, t1, 10, 14, focal, O.novel_poses(1)[0], white_bkgd=False, lindisp=True)
    assert np.abs(rgbs[0].reshape(-1, 3) - ref['rgb_map'].numpy()).max() <= 1e-4


def test_checkpoint_saved_from_cuda_dataparallel(pkg, tmp_path):
    """What the reference's training run writes (main.py:482-509, 1516-1542; helpers:408-425): the state_dict of an
    nn.DataParallel-wrapped model living on cuda:0 -- `module.` prefixes, CUDA-device tensors -- saved by a real
    torch.save.  A fresh process (the CLI) loads it through load_checkpoint and renders; compared with the CPU oracle."""
    import torch.nn as nn
    n_block = 3

    class Res(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = nn.Sequential(nn.Linear(256, 256), nn.ReLU(True), nn.Linear(256, 256))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.head = nn.Sequential(nn.Linear(1008, 256), nn.ReLU(True))
            self.body = nn.Sequential(*[Res() for _ in range(n_block)])
            self.tail = nn.Sequential(nn.Linear(256, 3), nn.Sigmoid())

    sd = O.make_r2l_state(seed=4, netdepth=2 + 2 * n_block)
    net = Net()
    net.load_state_dict(sd)
    dp = nn.DataParallel(net.cuda())
    saved = dp.state_dict()
    assert all(k.startswith('module.') and v.is_cuda for k, v in saved.items())
    ck = str(tmp_path / 'dp_cuda.tar')
    torch.save({'global_step': 123, 'best_psnr': 30.1, 'best_psnr_step': 100, 'network_fn_state_dict': saved,
                'optimizer_state_dict': {'state': {}, 'param_groups': []}}, ck)
    out = str(tmp_path / 'out')
    run_main(['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
              '--netdepth', str(2 + 2 * n_block), '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp',
              '--pretrained_ckpt', ck, '--render_only', '--render_test', '--synthetic_poses', '2', '--H', '32', '--outdir', out])
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    H = 16  # half_res in the config
    assert rgbs.shape == (2, H, H, 3)
    poses = O.novel_poses(2)
    for i in range(2):
        ref = O.r2l_render(sd, H, H, O.focal_from_angle(32) / 2., poses[i]).view(H, H, 3).numpy()
        assert np.abs(rgbs[i] - ref).max() <= 1e-4


def test_cli_explicit_split(pkg, tmp_path):
    """--precision fp16_split --split_block k taken literally (no probe, no watch: an explicit mode keeps itself); k = n_block renders
    what --precision fp16x3_asm renders, bit for bit; both inside the contract"""
    from efficient_nerf_amd import frontend as fe
    nb, H = 4, 32
    sd = O.make_r2l_state(seed=8, netdepth=2 + 2 * nb)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    base = ['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256', '--netdepth', str(2 + 2 * nb),
            '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--pretrained_ckpt', ck, '--render_only', '--synthetic_poses', '1', '--H', '64']
    outs = {}
    for tag, extra in (('s1', ['--precision', 'fp16_split', '--split_block', '1']), ('s4', ['--precision', 'fp16_split', '--split_block', str(nb)]),
                       ('x3', ['--precision', 'fp16x3_asm'])):
        log = run_main(base + extra + ['--outdir', str(tmp_path / tag)])
        outs[tag] = np.load(tmp_path / tag / 'rgbs.npy')
        if tag != 'x3':
            assert f'[precision] fp16_split: blocks [0, {extra[-1]}) in three fp16 passes' in log and 'rgb watch' not in log, log
    ref = O.r2l_render(sd, H, H, O.focal_from_angle(64) / 2., O.novel_poses(1)[0]).view(H, H, 3).numpy()
    assert np.array_equal(outs['s4'], outs['x3'])
    assert np.abs(outs['s1'][0] - ref).max() <= 1e-4 and 0 < np.abs(outs['s1'] - outs['x3']).max() <= 5e-5


def test_cli_rejects_unsupported(pkg, tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '--model_name', 'R2L'], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode != 0 and 'render_only' in (r.stdout + r.stderr)


def test_cli_precision_auto(pkg, tmp_path):
    """--precision auto measures the checkpoint's activation ranges on a probe render and says what it chose; the same
    checkpoint through the poses path and the given-rays path, a stress checkpoint (body weights x 1.3, 43 blocks)
    goes behind the whole-network rungs (fp16_split with the split it measured, or fp16x3_asm); every render stays within 1e-4 of the oracle"""
    from efficient_nerf_amd import frontend as fe
    H = 32
    focal = O.focal_from_angle(64) / 2.
    poses = O.novel_poses(1)
    for tag, gain, want in (('std', 1.0, 'fp16_fp8'), ('stress', 1.3, 'fp16x3_asm')):
        sd = O.make_r2l_state(seed=0)
        for k in sd:
            if k.startswith('body.') and k.endswith('weight'):
                sd[k] = sd[k] * gain
        ck = str(tmp_path / f'{tag}.tar')
        fe.save_checkpoint(ck, sd)
        base = ['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
                '--netdepth', '88', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--pretrained_ckpt', ck,
                '--render_only', '--synthetic_poses', '1', '--H', '64', '--precision', 'auto']
        log = run_main(base + ['--outdir', str(tmp_path / tag)])
        said = log.split('[precision] auto')[1].splitlines()[0].rstrip()
        assert said.endswith('-> ' + want) or (want == 'fp16x3_asm' and said.endswith(('-> fp16_split', '-> fp16_split8')) and 'leading blocks in three passes' in log), log
        ref = O.r2l_render(sd, H, H, focal, poses[0]).view(H, H, 3).numpy()
        a = np.load(tmp_path / tag / 'rgbs.npy')
        assert np.abs(a[0] - ref).max() <= 1e-4
        if tag == 'std':
            ro, rd = O.rays_from_dirs(O.camera_dirs(H, H, focal), poses[0][:3, :4])
            rays = str(tmp_path / 'rays.pt')
            torch.save({'all_rays_o': ro.reshape(1, -1, 3), 'all_rays_d': rd.reshape(1, -1, 3)}, rays)
            log = run_main(base + ['--outdir', str(tmp_path / 'g'), '--given_render_path_rays', rays])
            assert '-> fp16_fp8' in log
            assert np.abs(np.load(tmp_path / 'g' / 'rgbs.npy')[0] - ref).max() <= 1e-4


def test_r2l_render_only_cli_two_ranks(pkg, tmp_path):
    """main.py --render_only under torch.distributed.run with two ranks (gloo between them, both on this GPU): rows
    sharded 17 / 16 (H = 33 is ragged), fp16_fp8 with the exponents agreed between the ranks, one gather per frame;
    rank 0 writes the frames.  Every frame within 1e-4 of the oracle."""
    import socket
    from efficient_nerf_amd import frontend as fe
    sd = O.make_r2l_state(seed=12, netdepth=14)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    out = str(tmp_path / 'out')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, R2L_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'main.py'),
                        '--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
                        '--netdepth', '14', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--pretrained_ckpt', ck,
                        '--render_only', '--synthetic_poses', '3', '--H', '66', '--precision', 'fp16_fp8', '--outdir', out],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'on 2 GPU(s)' in r.stdout
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    H = 33
    assert rgbs.shape == (3, H, H, 3)
    focal = O.focal_from_angle(66) / 2.
    for i, c2w in enumerate(O.novel_poses(3)):
        ref = O.r2l_render(sd, H, H, focal, c2w).view(H, H, 3).numpy()
        assert np.abs(rgbs[i] - ref).max() <= 1e-4, i


def test_cli_auto_with_a_layer_the_generated_kernels_cannot_pack(pkg, tmp_path):
    """`--precision auto` (the CLI default) on a checkpoint with a layer of max|w| = 2^8: the weights load in
    fp16x3, choose_precision finds the generated modes refuse them and stays there, saying why -- the checkpoint renders,
    inside the contract, instead of failing in r2l_load_weights"""
    from efficient_nerf_amd import frontend as fe
    sd = O.make_r2l_state(seed=3, netdepth=10)
    sd['body.1.body.0.weight'] = sd['body.1.body.0.weight'] * 4096.
    sd['body.1.body.0.bias'] = sd['body.1.body.0.bias'] * 4096.
    sd['body.1.body.2.weight'] = sd['body.1.body.2.weight'] / 4096.
    ck = str(tmp_path / 'odd.tar')
    fe.save_checkpoint(ck, sd)
    out = str(tmp_path / 'out')
    log = run_main(['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
                    '--netdepth', '10', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--pretrained_ckpt', ck,
                    '--render_only', '--synthetic_poses', '1', '--H', '48', '--outdir', out])
    assert '[precision] auto' in log and 'outside the range' in log and '-> fp16x3' in log, log
    H = 24
    ref = O.r2l_render(sd, H, H, O.focal_from_angle(48) / 2., O.novel_poses(1)[0]).view(H, H, 3).numpy()
    assert np.abs(np.load(os.path.join(out, 'rgbs.npy'))[0] - ref).max() <= 1e-4


def test_render_path_batches_frames_per_collective_two_ranks(pkg, tmp_path):
    """`torchrun main.py` runs the loop bench.py times.  Two ranks (gloo between them, both on this GPU),
    H = 33 (ragged 17 / 16 rows), 5 frames, `--precision auto`: every rank renders its row shard of BOTH frames of a batch in
    one launch, one collective and one range check per batch (3 batches for 5 frames: the CLI prints the counts), PNGs
    come from the writer threads, every frame within 1e-4 of the oracle."""
    import re
    import socket
    from efficient_nerf_amd import frontend as fe
    sd = O.make_r2l_state(seed=12, netdepth=14)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    out = str(tmp_path / 'out')
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, R2L_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'main.py'),
                        '--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
                        '--netdepth', '14', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--pretrained_ckpt', ck,
                        '--render_only', '--synthetic_poses', '5', '--H', '66', '--outdir', out],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r'Render loop: 5 view\(s\) 33x33 on 2 GPU\(s\) .* \((\d+) batch\(es\) of (\d+) frame\(s\), (\d+) collective\(s\), (\d+) re-render',
                  r.stdout)
    assert m, r.stdout[-2000:]
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (3, 2, 3), m.group(0)
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    H = 33
    assert rgbs.shape == (5, H, H, 3)
    focal = O.focal_from_angle(66) / 2.
    for i, c2w in enumerate(O.novel_poses(5)):
        ref = O.r2l_render(sd, H, H, focal, c2w).view(H, H, 3).numpy()
        assert np.abs(rgbs[i] - ref).max() <= 1e-4, i
        assert os.path.getsize(os.path.join(out, f'{i:03d}.png')) > 100


def test_main_starts_its_own_ranks_and_watches_a_teacher_path(pkg, tmp_path):
    """`python main.py --gpus 2 ...` without torchrun: the entry script starts both ranks (launch.py; gloo between
    them, both on this GPU).  A teacher checkpoint, five frames, `--precision auto`: probes on three poses of the path, the watch's spot
    checks agreed between the ranks (frame 0: every rank re-renders 2,048 rays of its own rows in fp16x3), frames within 1e-4 of the oracle."""
    from efficient_nerf_amd import frontend as fe
    t0, t1 = O.make_teacher_state(1), O.make_teacher_state(2)
    ck = str(tmp_path / 'nerf.tar')
    fe.save_checkpoint(ck, t0, t1)
    out = str(tmp_path / 'out')
    env = dict(os.environ, R2L_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py'), '--gpus', '2', '--launch_timeout', '500', '--model_name', 'nerf',
                        '--config', 'configs/lego.txt', '--pretrained_ckpt', ck, '--render_only', '--synthetic_poses', '5', '--H', '34',
                        '--outdir', out], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'of each of 3 probe frame(s)' in r.stdout and '-> fp16x1' in r.stdout, r.stdout[-1500:]
    assert 'on 2 GPU(s)' in r.stdout and '[precision] watch: 1 spot check(s)' in r.stdout and '0 fallback(s)' in r.stdout, r.stdout[-1500:]
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    H = 17
    assert rgbs.shape == (5, H, H, 3)
    focal = O.focal_from_angle(34) / 2.
    for i, c2w in enumerate(O.novel_poses(5)):
        ref = O.teacher_render(t0, t1, H, H, focal, c2w, white_bkgd=True)['rgb_map'].view(H, H, 3).numpy()
        assert np.abs(rgbs[i] - ref).max() <= 1e-4, i


def test_cli_activation_variants(pkg, tmp_path):
    """`--act lrelu --trial.inact lrelu --trial.outact relu` (model/nerf_raybased.py:468-476, 497-522) through the command line with
    its default --precision auto: the generated modes refuse the network, auto says so and renders in fp16x3 within the contract"""
    from efficient_nerf_amd import frontend as fe
    sd = O.make_r2l_state(seed=9, netdepth=8)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    out = str(tmp_path / 'out')
    log = run_main(['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
                    '--netdepth', '8', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--act', 'lrelu', '--trial.inact', 'lrelu',
                    '--trial.outact', 'relu', '--trial.res_scale', '0.5', '--pretrained_ckpt', ck, '--render_only', '--synthetic_poses', '1',
                    '--H', '48', '--outdir', out])
    assert '[precision] auto' in log and 'relu / relu / none' in log and '-> fp16x3' in log, log
    H = 24
    c2w = O.novel_poses(1)[0]
    pts = O.sample_test(O.camera_dirs(H, H, O.focal_from_angle(48) / 2.), O.sampler_z_vals(16, 2., 6.), c2w[:3, :4])
    ref = O.r2l_forward(sd, O.positional_embed(pts, 10), res_scale=0.5, act='lrelu', inact='lrelu', outact='relu').view(H, H, 3).numpy()
    assert np.abs(np.load(os.path.join(out, 'rgbs.npy'))[0] - ref).max() <= 1e-4


def test_cli_plain_mlp_body(pkg, tmp_path):
    """`--trial.body_arch mlp` (the constructor's default architecture, model/nerf_raybased.py:515-518) through the command line: the
    state_dict keys are body.{0,2,4,...}; rendered in fp16x3, within the contract of the oracle's restatement of that network"""
    from efficient_nerf_amd import frontend as fe
    sd = O.make_r2l_mlp_state(seed=11, netdepth=8)
    ck = str(tmp_path / 'mlp.tar')
    fe.save_checkpoint(ck, sd)
    out = str(tmp_path / 'out')
    log = run_main(['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '16', '--netwidth', '256',
                    '--netdepth', '8', '--use_residual', '--trial.ON', '--trial.body_arch', 'mlp', '--pretrained_ckpt', ck, '--render_only',
                    '--synthetic_poses', '1', '--H', '48', '--outdir', out])
    assert '-> fp16x3' in log, log
    H = 24
    pts = O.sample_test(O.camera_dirs(H, H, O.focal_from_angle(48) / 2.), O.sampler_z_vals(16, 2., 6.), O.novel_poses(1)[0][:3, :4])
    ref = O.r2l_forward_mlp(sd, O.positional_embed(pts, 10)).view(H, H, 3).numpy()
    assert np.abs(np.load(os.path.join(out, 'rgbs.npy'))[0] - ref).max() <= 1e-4


def test_cli_renders_shapes_outside_the_fused_kernels(pkg, tmp_path):
    """The reference's command line with networks the fused kernels are not built for (another width / sample count / number of
    frequencies, three Linear layers per block; a 4 x 128 teacher with a 6 x 96 fine network): `--precision auto` says it takes the
    generic fp32 layer path and the frames are within the contract of the CPU oracle; an explicit fused precision is refused."""
    from efficient_nerf_amd import frontend as fe
    H = 24
    focal = O.focal_from_angle(48) / 2.
    trial = dict(body_arch='resmlp', n_block=3, n_learnable=3, res_scale=0.5, inact='lrelu', outact='none')
    sd = O.make_v3_2_state(11, 12, 96, 3 * 8 * 13, '', 'relu', trial)
    ck = str(tmp_path / 'r2l.tar')
    fe.save_checkpoint(ck, sd)
    out = str(tmp_path / 'out')
    flags = ['--model_name', 'R2L', '--config', 'configs/lego_noview.txt', '--n_sample_per_ray', '8', '--multires', '6', '--netwidth', '96',
             '--netdepth', '12', '--use_residual', '--trial.ON', '--trial.body_arch', 'resmlp', '--trial.n_block', '3', '--trial.n_learnable', '3',
             '--trial.res_scale', '0.5', '--trial.inact', 'lrelu', '--pretrained_ckpt', ck, '--render_only', '--render_test', '--testskip', '1',
             '--synthetic_poses', '2', '--H', '48', '--outdir', out]
    log = run_main(flags)
    assert 'generic fp32 layer path' in log, log[-1500:]
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    assert rgbs.shape == (2, H, H, 3)
    for i, c2w in enumerate(O.novel_poses(2)):
        pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(8, 2., 6.), c2w[:3, :4])
        ref = O.v3_2_forward(sd, O.positional_embed(pts, 6), 12, 'relu', True, trial).view(H, H, 3).numpy()
        assert np.abs(rgbs[i] - ref).max() <= 1e-4
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'main.py')] + flags + ['--precision', 'fp16_fp8'], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0 and 'generic path' in (r.stdout + r.stderr)
    # the teacher
    t0 = O.make_nerf_state(21, 4, 128, 63, 27, 5, (4,), True)
    t1 = O.make_nerf_state(22, 6, 96, 63, 27, 5, (4,), True)
    ck = str(tmp_path / 'nerf.tar')
    fe.save_checkpoint(ck, t0, t1)
    out = str(tmp_path / 'out_nerf')
    log = run_main(['--model_name', 'nerf', '--config', 'configs/lego.txt', '--netdepth', '4', '--netwidth', '128', '--netdepth_fine', '6',
                    '--netwidth_fine', '96', '--pretrained_ckpt', ck, '--render_only', '--render_test', '--testskip', '1', '--synthetic_poses', '1',
                    '--H', '16', '--outdir', out])
    assert 'generic fp32 layer path' in log, log[-1500:]
    rgbs = np.load(os.path.join(out, 'rgbs.npy'))
    assert rgbs.shape == (1, 8, 8, 3)
    ro, rd = O.get_rays(8, 8, O.focal_from_angle(16) / 2., O.novel_poses(1)[0][:3, :4])
    ref = O.render_rays_generic(t0, t1, ro.reshape(-1, 3).float(), rd.reshape(-1, 3).float(), 2., 6., 64, 128, True)
    assert np.abs(rgbs[0].reshape(-1, 3) - ref['rgb_map'].numpy()).max() <= 1e-4
