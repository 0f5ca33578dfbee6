#!/usr/bin/env python
"""Whole-frame parity of the trained-like teacher: every ray of three 400 x 400 frames, not a strided sample.

TEST INFRASTRUCTURE.  Two steps:
  --oracle   (CPU, no GPU): the fp32 CPU oracle (oracle.render_rays = main.py:624-756 restated) AND the same functions in float64
             (oracle.render_rays_taps(dtype=float64): the stand-in for exact arithmetic) on every ray of the three poses ->
             tests/golden/trained_like/teacher_whole_frame.npz (rgb / acc / depth of the fp32 oracle, rgb of float64, the number of
             sample_pdf index flips between the two per ray).  ~10 min per pose on 8 threads.
  (default)  (GPU): renders the frames in fp16x3_asm and fp16x3 (and any --modes), counts rays beyond 1e-4 of the fp32 oracle and
             classifies every ray where any pair (HIP mode, fp32 oracle) differs by more than 5e-5 (oracle.whole_frame.classify).
Output: a text report (profiles/r06_teacher_whole_frame.txt) + JSON."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402
from oracle import whole_frame as WF  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--oracle', action='store_true')
    ap.add_argument('--threads', type=int, default=8)
    ap.add_argument('--modes', default='fp16x3_asm,fp16x3')
    ap.add_argument('--poses', default='0,1,2')
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'teacher_whole_frame'))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    sds = WF.load_teacher()
    poses = [int(p) for p in a.poses.split(',')]
    if a.oracle:
        WF.make_fixture(sds, log=lambda s: print(s, flush=True))
        return
    import _pkg
    _pkg.load()
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    fx = WF.load_fixture()
    H = WF.H
    eng = NeRFEngine(H, H, WF.focal(), precision=PRECISIONS['fp16x3']).load_state_dicts(*sds)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    lines, summary = [], {}
    say = lambda s: (print(s, flush=True), lines.append(s))
    say(f'# whole-frame parity of the trained-like teacher: {H} x {H}, every ray; poses {[WF.POSES[p] for p in poses]}')
    say(f'# fp32 oracle = oracle.render_rays on the CPU ({fx["host"]}); float64 = the same functions in double on the same float32 inputs')
    for pi in poses:
        st = WF.fixture_stats(fx, pi)
        say(f'pose {pi}: fp32 oracle vs float64 over {H * H} rays: rgb L_inf {st["linf"]:.2e}; rays > 1e-5: {st["n_1e5"]}, > 5e-5: {st["n_5e5"]}, '
            f'> 1e-4: {st["n_1e4"]}, > 1e-3: {st["n_1e3"]}; rays with a searchsorted index flip fp32 vs float64: {st["n_flip"]}')
    for mode in a.modes.split(','):
        eng.set_precision(PRECISIONS[mode])
        tot = dict(rays=0, n_gt=0, n_expl=0, worst_unexpl=0.0, worst_all=0.0, classes={})
        for pi in poses:
            t0 = time.time()
            r = WF.classify_frame(eng, sds, fx, pi, log=say, label=f'{mode} pose {pi}')
            tot['rays'] += r['rays']
            tot['n_gt'] += r['n_gt_1e-4_vs_fp32_oracle']
            tot['n_expl'] += r['n_explained_by_f64']
            tot['worst_unexpl'] = max(tot['worst_unexpl'], r['worst_unexplained'])
            tot['worst_all'] = max(tot['worst_all'], r['linf_vs_fp32_oracle'])
            for k, v in r['classes'].items():
                tot['classes'][k] = tot['classes'].get(k, 0) + v
            say(f'{mode} pose {pi}: {json.dumps({k: v for k, v in r.items() if k != "detail"})}  ({time.time() - t0:.0f} s)')
        summary[mode] = tot
        say(f'== {mode}: {tot["rays"]} rays, L_inf vs fp32 oracle {tot["worst_all"]:.2e}; > 1e-4: {tot["n_gt"]}, of which explained by the float64 '
            f'evaluation {tot["n_expl"]}; classes of the rays examined {tot["classes"]}; worst unexplained {tot["worst_unexpl"]:.2e}')
    # the two fp32-grade modes against each other (NeRFEngine._x3_pair: what `auto` and the watch compare): coarse maps on every ray,
    # fine maps on the rays whose fine samples did not move
    if 'fp16x3_asm' in a.modes and 'fp16x3' in a.modes:
        from efficient_nerf_amd.teacher import get_rays
        for pi in poses:
            ro, rd = (t.reshape(-1, 3) for t in get_rays(H, H, WF.focal(), WF.pose(pi)[:3, :4], device=eng.device))
            eng.set_precision(PRECISIONS['fp16x3'])
            ref = {k: v.clone() for k, v in eng.render_rays(ro, rd, extras=True).items()}
            eng.set_precision(PRECISIONS['fp16x3_asm'])
            d, good = eng._x3_pair(ro, rd, ref)
            allr = (eng.render_rays(ro, rd)['rgb_map'] - ref['rgb_map']).abs().max(-1)[0]
            say(f'pose {pi}: fp16x3_asm against fp16x3 over the whole frame, stage by stage (coarse maps; fine pass at fp16x3\'s sample positions): '
                f'{json.dumps({k: float("%.3g" % v) for k, v in d.items()})} (limit {eng.AUTO_MAX_DIFF_X3ASM:g}: {"ok" if good else "MISS"}); the two full '
                f'renders: rgb {float(allr.max()):.2e}, rays > 1e-4: {int((allr > 1e-4).sum())}')
            summary.setdefault('x3_pair', {})[pi] = d
    eng.close()
    with open(a.out + '.txt', 'w') as f:
        f.write('\n'.join(lines) + '\n')
    with open(a.out + '.json', 'w') as f:
        json.dump(summary, f, indent=1)


if __name__ == '__main__':
    main()
