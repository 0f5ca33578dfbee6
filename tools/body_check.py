"""GPU check of the hand-scheduled body kernel: (1) r2l_debug_body against a float64 evaluation of the ResMLP
blocks on random register images, bitwise repeatability; (2) the full FP16_FP8 pipeline against the CPU
oracle; (3) timing of the 800x800 frame against the round-1 fused kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import _pkg
_pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X3, PREC_FP16_FP8
from oracle import r2l_oracle as O


def image_to_rays(x):  # [T,4,32,64,4] -> [T*128, 256]; register image of csrc/r2l_common.h (32x32 MFMA shapes)
    T = x.shape[0]
    x = x.reshape(T, 4, 8, 4, 2, 32, 4)           # tile, wave, u, g, h, ray, i: feature 32u + 8g + 4h + i
    x = x.permute(0, 1, 5, 2, 3, 4, 6)            # tile, wave, ray, u, g, h, i
    return x.reshape(T * 128, 256)


def rays_to_image(r):
    T = r.shape[0] // 128
    x = r.reshape(T, 4, 32, 8, 4, 2, 4).permute(0, 1, 3, 4, 5, 2, 6)
    return x.reshape(T, 4, 32, 64, 4).contiguous()


def body_check(nb, n_tiles, seed=0):
    sd = O.make_r2l_state(seed=seed, netdepth=2 + 2 * nb)
    eng = R2LEngine(64, 64, O.focal_from_angle(64), n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    g = torch.Generator().manual_seed(seed + 1)
    xr = torch.relu(torch.randn(n_tiles * 128, 256, generator=g))
    S = 16.0
    xin = rays_to_image(xr * S).cuda()
    assert torch.equal(image_to_rays(xin.cpu()), xr * S)
    out = eng.debug_body(xin)
    torch.cuda.synchronize()
    out2 = eng.debug_body(xin)
    torch.cuda.synchronize()
    same = torch.equal(out, out2)
    x = xr.double()
    Bsum = torch.zeros(256, dtype=torch.float64)
    for i in range(nb):
        W1, b1 = sd[f'body.{i}.body.0.weight'].double(), sd[f'body.{i}.body.0.bias'].double()
        W2, b2 = sd[f'body.{i}.body.2.weight'].double(), sd[f'body.{i}.body.2.bias'].double()
        x = x + torch.relu(x @ W1.T + b1) @ W2.T + b2
        Bsum += b2
    got = image_to_rays(out.cpu()).double() / S + Bsum
    err = (got - x).abs().max().item()
    print(f'body nb={nb} tiles={n_tiles}: L_inf {err:.3e} (max|x| {x.abs().max().item():.2f}) repeatable={same} '
          f'finite={bool(torch.isfinite(out).all())}', flush=True)
    return err, same


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    torch.cuda.set_device(0)
    if what in ('all', 'body'):
        for nb, nt in ((1, 1), (1, 3), (2, 300), (5, 700), (43, 520)):
            body_check(nb, nt)
    if what in ('all', 'full'):
        for H, nb in ((64, 3), (200, 43)):
            focal = O.focal_from_angle(H)
            sd = O.make_r2l_state(seed=0, netdepth=2 + 2 * nb)
            c2w = O.pose_spherical(30., -30., 4.)
            ref = O.r2l_render(sd, H, H, focal, c2w)
            for name, prec in (('fp16x3', PREC_FP16X3), ('fp16_fp8', PREC_FP16_FP8)):
                eng = R2LEngine(H, H, focal, n_block=nb, precision=prec).load_state_dict(sd)
                rgb = eng.render(c2w).cpu()
                print(f'full H={H} nb={nb} {name}: L_inf vs oracle {(rgb - ref).abs().max().item():.3e}', flush=True)
    if what in ('all', 'time'):
        H, nb = 800, 43
        focal = O.focal_from_angle(H)
        sd = O.make_r2l_state(seed=0, netdepth=2 + 2 * nb)
        poses = O.novel_poses(8)[:, :3, :4].contiguous().cuda()
        outs = {}
        for name, prec in (('fp16x3', PREC_FP16X3), ('fp16_fp8', PREC_FP16_FP8)):
            eng = R2LEngine(H, H, focal, n_block=nb, precision=prec).load_state_dict(sd)
            for _ in range(3):
                eng.render_batch(poses[0:1])
            torch.cuda.synchronize()
            eng.timing(True)
            t0 = time.time()
            n = 10
            for i in range(n):
                out = eng.render_batch(poses[i % 8:i % 8 + 1])
            torch.cuda.synchronize()
            dt = (time.time() - t0) / n
            kt, kn = eng.kernel_time_ms()
            outs[name] = eng.render_batch(poses[0:1]).cpu()
            print(f'time {name}: {dt * 1e3:.3f} ms/frame = {H * H / dt:.3e} rays/s; timed kernel {kt / max(kn, 1):.3f} ms '
                  f'({eng.kernel_flops_per_ray * H * H / (kt / max(kn, 1) * 1e-3) / 2.5e15:.3f} of 2.5 PF)', flush=True)
        print('fp16x3 vs fp16_fp8 L_inf', (outs['fp16x3'] - outs['fp16_fp8']).abs().max().item())


if __name__ == '__main__':
    main()
