#!/usr/bin/env python
"""Config 5 over MANY save groups on the GPU: the pipeline's buffers cycle (two device row buffers, two pinned host buffers, the
planner's queue, the writer threads) -- 600 random poses at 400x400 in groups of 100 twice, the two directories compared file by file,
and the first group compared with a 100-pose run (a group's shards do not depend on what follows).
    python tools/create_data_soak.py [n_pose] [H]        (through gpurun; ~2 minutes at the defaults)
CD_TEACHER=trained takes the trained-like teacher of tests/golden/trained_like instead of the synthetic one, CD_PREC=auto lets
create_data.choose_precision_for_rand measure the mode (the per-group watch runs as in `python create_data.py`)."""
import hashlib
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import _pkg  # noqa: E402

_pkg.load()
from efficient_nerf_amd import NeRFEngine, PRECISIONS  # noqa: E402
from efficient_nerf_amd.create_data import RandStream, create_rand, choose_precision_for_rand  # noqa: E402
from oracle import r2l_oracle as O  # noqa: E402
from oracle import whole_frame as WF  # noqa: E402

n_pose = int(sys.argv[1]) if len(sys.argv) > 1 else 600
H = int(sys.argv[2]) if len(sys.argv) > 2 else 400
focal = O.focal_from_angle(H)
prec = os.environ.get("CD_PREC", "fp16x1")
sds = (WF.load_teacher(os.environ["CD_TEACHER_DIR"]) if os.environ.get("CD_TEACHER_DIR") else      # any directory with teacher_{coarse,fine}.npz
       WF.load_teacher() if os.environ.get("CD_TEACHER") == "trained" else (O.make_teacher_state(1), O.make_teacher_state(2)))
eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3' if prec == 'auto' else prec]).load_state_dicts(*sds)
if prec == 'auto':
    prec = choose_precision_for_rand(eng, H, H, focal)
    print(f'auto -> {prec}: {eng.auto_diffs}', flush=True)


def digest(d):
    return {n: hashlib.sha256(open(os.path.join(d, n), 'rb').read()).hexdigest() for n in sorted(os.listdir(d)) if n.endswith('.npy')}


res = {}
for tag, n in (('a', n_pose), ('b', n_pose), ('one', 100)):
    d = '/tmp/r2l_soak_' + tag
    shutil.rmtree(d, ignore_errors=True)
    tm = {}
    t0 = time.perf_counter()
    k = create_rand(eng, H, H, focal, n, d, i_save=100, split_size=4096, stream=RandStream(), log=lambda *a, **kw: None, timings=tm)
    dt = time.perf_counter() - t0
    res[tag] = digest(d)
    w = tm.get('watch') or {}
    print(f'{tag}: {prec} -> {eng.precision_name}, watch {w.get("checks")} checks, fallbacks {w.get("fallbacks")}, worst {w.get("worst")}', flush=True)
    print(f'{tag}: {n} poses, {k} shards, {dt:.2f} s = {n / dt:.2f} poses/s; MLP kernels {tm["mlp_kernel_ms"] / 1e3:.2f} s '
          f'({tm["mlp_kernel_ms"] / 1e3 / dt:.3f} of the wall clock), tail {tm["tail_s"]:.2f} s, planner {tm["permutation_s"]:.2f} s, '
          f'writers {tm["writer_busy_s"]:.2f} s', flush=True)
    shutil.rmtree(d, ignore_errors=True)
same = res['a'] == res['b']
first = all(res['a'].get(k) == v for k, v in res['one'].items())
print(f'{len(res["a"])} shards: the two runs are {"byte-identical" if same else "DIFFERENT"}; the first group equals the 100-pose run: {first}')
sys.exit(0 if same and first and len(res['a']) == n_pose // 100 * 3906 else 1)
