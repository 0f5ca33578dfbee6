#!/usr/bin/env python
"""Development probe (GPU box): kernel time of the fused R2L kernel vs depth / precision,
to split head (embedding + Linear 1008x256) from body (per ResMLP block) cost."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X1, PREC_FP16X3  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402

H = W = int(os.environ.get('PROBE_H', 800))
REP = int(os.environ.get('PROBE_REP', 5))


def time_engine(n_block, prec):
    sd = O.make_r2l_state(seed=0, netdepth=2 + 2 * n_block)
    eng = R2LEngine(H, W, O.focal_from_angle(W), n_block=n_block, precision=prec).load_state_dict(sd)
    poses = O.novel_poses(8)[:, :3, :4].contiguous().cuda()
    for i in range(2):
        eng.render_batch(poses[i:i + 1])
    torch.cuda.synchronize()
    eng.timing(True)
    eng.kernel_time_ms(reset=True)
    for i in range(REP):
        eng.render_batch(poses[i % 8:i % 8 + 1])
    ms, n = eng.kernel_time_ms(reset=True)
    eng.close()
    return ms / n


if __name__ == '__main__':
    blocks = [int(x) for x in os.environ.get('PROBE_BLOCKS', '0,11,43').split(',')]
    modes = os.environ.get('PROBE_MODES', 'fp16x3,fp16_fp8,fp16x1').split(',')
    for prec, name in ((PREC_FP16X3, 'fp16x3'), (2, 'fp16_fp8'), (PREC_FP16X1, 'fp16x1')):
        if name not in modes:
            continue
        res = {nb: time_engine(nb, prec) for nb in blocks}
        line = ' '.join(f'nb={nb}:{ms:.3f}ms' for nb, ms in res.items())
        nb0, nb1 = blocks[0], blocks[-1]
        per_block = (res[nb1] - res[nb0]) / max(nb1 - nb0, 1)
        rays = H * W
        passes = {'fp16x3': 3, 'fp16_fp8': 2, 'fp16x1': 1}[name]
        body_tf = 2 * 2 * 65536 * rays / (per_block * 1e-3) / 1e12 * passes
        head_tf = 2 * (1008 * 256 + 768) * rays / (res[nb0] * 1e-3) / 1e12 * passes if nb0 == 0 else float('nan')
        print(f'{name} {H}x{W}: {line} | per-block {per_block*1e3:.1f} us (executed {body_tf:.0f} TF/s) | '
              f'head+tail executed {head_tf:.0f} TF/s | rays/s @nb={nb1}: {rays/res[nb1]*1e3:.3e}', flush=True)
