"""Soak of one precision mode on the GPU box: S_REPS renders of the same frame must be bitwise equal (S_PREC: a name of
efficient_nerf_amd.PRECISIONS; S_GUARD: range-guard period of the modes that have one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import R2LEngine, PRECISIONS
from oracle import r2l_oracle as O
H = int(os.environ.get('S_H', 800)); nb = int(os.environ.get('S_NB', 43)); reps = int(os.environ.get('S_REPS', 50))
name = {'x1': 'fp16x1', 'x3': 'fp16x3', 'mix': 'fp16_fp8'}.get(os.environ.get('S_PREC', 'x3'), os.environ.get('S_PREC', 'x3'))
rows = int(os.environ.get('S_ROWS', H))
sd = O.make_r2l_state(seed=0, netdepth=2 + 2 * nb)
eng = R2LEngine(H, H, O.focal_from_angle(H), n_block=nb, precision=PRECISIONS[name]).load_state_dict(sd)
poses = O.novel_poses(8)[:, :3, :4].contiguous().cuda()
if name in ('fp16_fp8', 'fp16_e4m3', 'fp16_split', 'fp16_split8'):
    if name.startswith('fp16_split'):          # the two-part modes borrow fp16_fp8's exponents; S_SPLIT: blocks in three passes in front
        eng.set_precision(PRECISIONS['fp16_fp8'])
    eng.calibrate_on(c2w=O.novel_poses(8)[0])
    if name.startswith('fp16_split'):
        eng.set_precision(PRECISIONS[name])
        eng.set_split_block(int(os.environ.get('S_SPLIT', nb // 2)))
    eng.set_guard_period(int(os.environ.get('S_GUARD', 8)))
ref = None
bad = 0
for i in range(reps):
    out = eng.render_batch(poses[0:1], rows=(0, rows))
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    elif not torch.equal(out, ref):
        bad += 1
        print('MISMATCH at rep', i, (out - ref).abs().max().item(), flush=True)
    if i % 500 == 499:
        print(f'  {name}: {i + 1} frames', flush=True)
print(f'stress {"ok" if not bad else "FAILED"} H={H} rows={rows} nb={nb} prec={name} reps={reps} mismatches={bad} '
      f'finite={bool(torch.isfinite(ref).all())}', flush=True)
sys.exit(1 if bad else 0)
