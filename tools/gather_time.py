"""Launch cost of the library's collective (r2l_gather_image: one grouped RCCL launch) with ONE rank on one GPU: what a
step of bench.py --gpus N pays besides its wire time.  F frames of 800x800x3 f32 per call, as a step at N = F renders.
Run through gpurun: python tools/gather_time.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import _pkg
_pkg.load()
from efficient_nerf_amd import dist as D

H = W = 800
for F in (1, 2, 4, 8):
    local = torch.rand(F, H * W, 3, device='cuda')
    for _ in range(5):
        D.gather_rows(local, H, W, 1, force_collective=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 200
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        D.gather_rows(local, H, W, 1, force_collective=True)
    e1.record()
    torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / n
    print('r2l_gather_image, 1 rank, %d frame(s) of %.2f MB: %.1f us per call on the stream (%.1f us host wall), %.0f GB/s device copy'
          % (F, H * W * 12 / 1e6, e0.elapsed_time(e1) / n * 1e3, host * 1e6, F * H * W * 12 / (e0.elapsed_time(e1) / n * 1e-3) / 1e9), flush=True)
