"""Times the FP16_FP8 path on one 800x800 frame (R2L W256D88): wall per frame and the body kernel by HIP events.
R2L_LIB_PATH selects a build variant (tools/build_variant.sh).  BT_FRAMES, BT_H, BT_NB, BT_PREC override defaults."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg; _pkg.load()
from efficient_nerf_amd import R2LEngine, PRECISIONS
from oracle import r2l_oracle as O

H = int(os.environ.get('BT_H', 800)); nb = int(os.environ.get('BT_NB', 43)); n = int(os.environ.get('BT_FRAMES', 12))
prec = PRECISIONS[os.environ.get('BT_PREC', 'fp16_fp8')]
sd = O.make_r2l_state(seed=0, netdepth=2 + 2 * nb)
eng = R2LEngine(H, H, O.focal_from_angle(H), n_block=nb, precision=prec).load_state_dict(sd)
poses = O.novel_poses(8)[:, :3, :4].contiguous().cuda()
for _ in range(3):
    eng.render_batch(poses[0:1])
torch.cuda.synchronize()
if prec in (PRECISIONS['fp16_fp8'], PRECISIONS['fp16_e4m3']):
    eng.set_guard_period(int(os.environ.get('BT_GUARD', 0)))   # 0: the plain body kernel only
    eng.render_batch(poses[0:1])
eng.timing(True)
ts = []
for i in range(n):
    t0 = time.time(); eng.render_batch(poses[i % 8:i % 8 + 1]); torch.cuda.synchronize(); ts.append(time.time() - t0)
kt, kn = eng.kernel_time_ms()
ts.sort()
print(f'{os.path.basename(os.environ.get("R2L_LIB_PATH", "default"))} {os.environ.get("BT_PREC", "fp16_fp8")}: frame median {ts[len(ts)//2]*1e3:.3f} ms min {ts[0]*1e3:.3f} ms '
      f'= {H*H/ts[len(ts)//2]:.3e} rays/s; timed kernel mean {kt/max(kn,1):.3f} ms', flush=True)
