"""rays/s when every frame's rgb is also copied to (pinned) host memory, as the reference's render_path does before it
writes images (main.py:331-346): the boundary itself hands over no host buffers but the 48-byte pose.
Run through gpurun: python tools/pcie_inclusive.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import r2l_oracle as O  # noqa: E402
import _pkg  # noqa: E402
_pkg.load()
from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine  # noqa: E402

H = 800
focal = O.focal_from_angle(H)
eng = R2LEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dict(O.make_r2l_state(seed=0))
poses = [O.pose_spherical(float(t), -30., 4.) for t in range(0, 360, 18)]
out = torch.empty((H * H, 3), device='cuda')
host = [torch.empty((H * H, 3), pin_memory=True) for _ in range(2)]
for p in poses[:3]:
    eng.render(p, out=out)
torch.cuda.synchronize()
for mode in ('device only', 'device + D2H copy (serial)', 'device + D2H copy (copy stream, double-buffered)'):
    copy_s = torch.cuda.Stream()
    outs = [torch.empty((H * H, 3), device='cuda') for _ in range(2)]
    ev = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for rep in range(3):
        for i, p in enumerate(poses):
            k = i & 1
            if mode.startswith('device only'):
                eng.render(p, out=out)
            elif 'serial' in mode:
                eng.render(p, out=out)
                host[0].copy_(out, non_blocking=True)
            else:
                torch.cuda.current_stream().wait_event(done[k])     # the copy of frame i-2 has left outs[k]
                eng.render(p, out=outs[k])
                ev[k].record()
                with torch.cuda.stream(copy_s):
                    copy_s.wait_event(ev[k])
                    host[k].copy_(outs[k], non_blocking=True)
                    done[k].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (3 * len(poses))
    print('%-50s %.3f ms/frame  %.3e rays/s' % (mode, dt * 1e3, H * H / dt))
