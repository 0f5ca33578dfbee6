import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import R2LEngine, PREC_FP16X1, PREC_FP16X3
from oracle import r2l_oracle as O
H = int(os.environ.get('S_H', 800)); nb = int(os.environ.get('S_NB', 43)); reps = int(os.environ.get('S_REPS', 50))
prec = {'x1': PREC_FP16X1, 'x3': PREC_FP16X3, 'mix': 2}[os.environ.get('S_PREC', 'x3')]
rows = int(os.environ.get('S_ROWS', H))
sd = O.make_r2l_state(seed=0, netdepth=2 + 2 * nb)
eng = R2LEngine(H, H, O.focal_from_angle(H), n_block=nb, precision=prec).load_state_dict(sd)
poses = O.novel_poses(8)[:, :3, :4].contiguous().cuda()
ref = None
for i in range(reps):
    out = eng.render_batch(poses[0:1], rows=(0, rows))
    torch.cuda.synchronize()
    if ref is None:
        ref = out.clone()
    elif not torch.equal(out, ref):
        print('MISMATCH at rep', i, (out - ref).abs().max().item(), flush=True)
print(f'stress ok H={H} rows={rows} nb={nb} prec={prec} reps={reps} finite={bool(torch.isfinite(ref).all())}', flush=True)
