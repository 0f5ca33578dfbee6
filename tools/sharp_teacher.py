#!/usr/bin/env python
"""The sharpest teacher of the trained-like family (tools/train_like.py --variant 2: thin bars, sigma ~ 276) through the round's final
code: what `auto` gives it, what the mixed rung misses by, and what the two exact savings of nerf_set_skip_rgb0 are worth on the rung
it keeps (three passes for both networks).  TEST INFRASTRUCTURE (fits with torch autograd, checks against the CPU oracle).

    python tools/sharp_teacher.py [--variant 2] [--teacher-steps 4000] [--dir gpurun_out/sharp_v2]      (through gpurun; ~8 minutes)

The fitted weights are kept under --dir (they travel back with gpurun_out/ and are reused by a second call); reports only are committed."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
import train_like as TL  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variant', type=int, default=2)
    ap.add_argument('--teacher-steps', type=int, default=4000)
    ap.add_argument('--dir', default=os.path.join(ROOT, 'gpurun_out', 'sharp_v2'))
    a = ap.parse_args()
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    from efficient_nerf_amd import create_data as CD
    TL.use_variant(a.variant)
    os.makedirs(a.dir, exist_ok=True)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)
    files = [os.path.join(a.dir, f'teacher_{n}.npz') for n in ('coarse', 'fine')]
    if all(os.path.exists(f) for f in files):
        sds = [TL.load_sd(f) for f in files]
        say(f'# teacher from {a.dir}')
    else:
        t0 = time.time()
        sds = TL.fit_teacher(a.teacher_steps, 2048, torch.device('cuda'), lambda m: print(m, flush=True))
        for f, sd in zip(files, sds):
            TL.save_sd(f, sd)
        say(f'# teacher: scene variant {a.variant}, {a.teacher_steps} steps, fitted in {time.time() - t0:.0f} s')
    H = TL.H_T
    focal = O.focal_from_angle(H)
    poses = [O.pose_spherical(30., -30., 4.), O.pose_spherical(150., -85., 4.), O.pose_spherical(-100., -5., 4.)]
    with torch.no_grad():
        eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3']).load_state_dicts(*sds)
        name = CD.choose_precision_for_rand(eng, H, H, focal)
        say(f'auto -> {name}; probe differences {json.dumps({k: float("%.3g" % v) for k, v in eng.auto_diffs.items()})}; limits: fp16_mix '
            f'{eng.AUTO_MAX_DIFF_MIX:g} against fp16x3_asm, fp16x3_asm {eng.AUTO_MAX_DIFF_X3ASM:g} against fp16x3 stage by stage')

        def timed(pose):
            eng.render(pose)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.render(pose)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 3 * 1e3
        res = {}
        for mode in ('fp16x3_asm', 'fp16_mix'):
            eng.set_precision(PRECISIONS[mode])
            rows = []
            for pi, pose in enumerate(poses):
                eng.set_skip_rgb0(False)
                full = {k: v.clone() for k, v in eng.render(pose, extras=True).items()}
                t_full = timed(pose)
                eng.set_skip_rgb0(True)
                got = eng.render(pose, extras=True)
                t_skip = timed(pose)
                same = all(torch.equal(got[k].view(torch.int32), full[k].view(torch.int32)) for k in got if k != 'raw')
                raw = full['raw'].reshape(-1, 4)
                n = raw.shape[0] // 128 * 128
                dead = float((~(raw[:n, 3] > 0).reshape(-1, 128).any(-1)).float().mean())
                rows.append(dict(pose=pi, ms_full=t_full, ms_skip=t_skip, maps_bitwise_equal=bool(same), dead_tiles=dead,
                                 sigma_max=float(raw[:, 3].max()), acc_lt_005=float((full['acc_map'] < .05).float().mean())))
                res.setdefault(mode + '_rgb', []).append(full['rgb_map'].clone())
            res[mode] = rows
            ms_f, ms_s = (sum(r[k] for r in rows) / 3 for k in ('ms_full', 'ms_skip'))
            frac = lambda ms: 303.82e6 * H * H / (ms * 1e-3) / 2.5e15
            say(f'{mode}: {ms_f:.1f} ms per 400 x 400 frame = {frac(ms_f):.3f} of the fp16 peak; with nerf_set_skip_rgb0 (coarse pass without its view '
                f'branch, fine pass with the second exit) {ms_s:.1f} ms = {frac(ms_s):.3f} ({(1 - ms_s / ms_f) * 100:.1f} % less); every map bitwise equal: '
                f'{all(r["maps_bitwise_equal"] for r in rows)}; fine-launch tiles without a positive density: '
                f'{", ".join("%.3f" % r["dead_tiles"] for r in rows)}; sigma max {max(r["sigma_max"] for r in rows):.0f}; rays with acc < 0.05: '
                f'{", ".join("%.2f" % r["acc_lt_005"] for r in rows)}')
        d = [float((x - y).abs().max()) for x, y in zip(res['fp16_mix_rgb'], res['fp16x3_asm_rgb'])]
        n5 = [int(((x - y).abs().max(-1)[0] > 5e-5).sum()) for x, y in zip(res['fp16_mix_rgb'], res['fp16x3_asm_rgb'])]
        say(f'fp16_mix against fp16x3_asm over the three whole frames: rgb L_inf {", ".join("%.2e" % v for v in d)}; rays beyond 5e-5: {n5} of {H * H} each')
        # the rung `auto` chose, against the CPU oracle on strided rays of each frame
        eng.set_precision(PRECISIONS[name])
        eng.set_skip_rgb0(True)
        idx = torch.arange(0, H * H, 157)
        worst = 0.0
        for pose in poses:
            ro, rd = O.get_rays(H, H, focal, pose[:3, :4])
            want = O.render_rays(sds[0], sds[1], ro.reshape(-1, 3)[idx].float(), rd.reshape(-1, 3)[idx].float(), white_bkgd=True)
            worst = max(worst, float((eng.render(pose)['rgb_map'].cpu()[idx] - want['rgb_map']).abs().max()))
        say(f'{name} (what auto chose, both savings on) against the CPU oracle on {len(idx)} strided rays x 3 poses: rgb L_inf {worst:.2e}')
        eng.close()
    with open(os.path.join(a.dir, 'report.txt'), 'w') as f:
        f.write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
