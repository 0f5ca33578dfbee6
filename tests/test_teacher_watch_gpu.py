"""GPU: the teacher's fast modes under watch.  `--precision auto` decides fp16x1 /
fp16_fp8 / fp16x3 once, on probes; NeRFEngine.spot_check re-renders a sample of the rays of what was rendered afterwards in
fp16x3 and step_down falls back one rung -- per save group in create_data (utils/create_data.py:812-872), every few frames in
render_path (main.py:272-350).  Also: the column tilings of the fp16x1 chain are bitwise equal (was tools/teacher_tile_stress.py)
and the tile count is a context field."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

SPIKE_GAIN = 64.0      # tools/teacher_view_spike.py: view-layer units that are dead for horizontal cameras, large for cameras looking down


def _spiked(seed):
    from teacher_view_spike import spiked_teacher
    return spiked_teacher(seed, SPIKE_GAIN)


def _rays(H, focal, pose, dev='cuda'):
    from efficient_nerf_amd import get_rays
    ro, rd = get_rays(H, H, focal, pose[:3, :4], device=dev)
    return ro.reshape(-1, 3), rd.reshape(-1, 3)


HORIZONTAL = (30., -5.)
TOP_DOWN = (150., -88.)


def test_spot_check_passes_where_the_probe_passed_and_catches_the_other_pose(pkg):
    """a teacher whose single-pass error depends on the camera: probed from the side `auto` takes fp16x1; a frame from above is
    outside the limit -- spot_check on 2,048 of ITS rays says so, step_down moves down the ladder until the check passes, and the
    frame rendered then is inside the contract against the CPU oracle"""
    from efficient_nerf_amd import NeRFEngine
    H = 128
    focal = O.focal_from_angle(H)
    sds = (_spiked(1), _spiked(2))
    eng = NeRFEngine(H, H, focal).load_state_dicts(*sds)
    side, top = O.pose_spherical(*HORIZONTAL, 4.), O.pose_spherical(*TOP_DOWN, 4.)
    name, diff = eng.choose_precision(*_rays(H, focal, side))
    print('probe from the side:', eng.auto_detail)
    assert name == 'fp16x1' and eng.precision_name == 'fp16x1'
    ro, rd = _rays(H, focal, side)
    ok, d = eng.spot_check(ro, rd, eng.render_rays(ro, rd))
    assert ok and d['rgb_map'] <= eng.AUTO_MAX_DIFF_X1, d
    ro, rd = _rays(H, focal, top)
    got = eng.render_rays(ro, rd)
    ok, d = eng.spot_check(ro, rd, got)
    print('frame from above in fp16x1:', d)
    assert not ok and d['rgb_map'] > eng.AUTO_MAX_DIFF_X1
    assert eng.precision_name == 'fp16x1'                  # a check changes nothing by itself
    rungs = []
    while not ok:
        rungs.append(eng.step_down())
        got = eng.render_rays(ro, rd)
        ok, d = eng.spot_check(ro, rd, got)
    print('after', rungs, d)
    assert rungs and rungs[-1] in ('fp16_fp8', 'fp16x3_asm') and eng.watch_fallbacks == len(rungs)
    idx = torch.arange(0, H * H, 37)
    want = O.render_rays(sds[0], sds[1], ro[idx].cpu(), rd[idx].cpu(), white_bkgd=True)['rgb_map']
    assert (got['rgb_map'][idx].cpu() - want).abs().max().item() <= 1e-4
    # probing the poses the job renders (what create_data / render_path now do) would not have chosen fp16x1 in the first place
    eng2 = NeRFEngine(H, H, focal).load_state_dicts(*sds)
    name2, _ = eng2.choose_precision([_rays(H, focal, side), _rays(H, focal, top)])
    assert name2 != 'fp16x1' and len(eng2.auto_detail['fp16x1']) == 2
    eng.close()
    eng2.close()


def test_create_rand_falls_back_inside_the_group_that_misses(pkg, tmp_path):
    """create_rand with an engine left in fp16x1 by a side-view probe: the first save group's first pose is checked against fp16x3,
    misses (the random poses look down), the engine steps down, the pose is rendered again -- the shards equal those of a run that
    was in the final mode from the start, bit for bit; `timings['watch']` records it.  watch=False keeps the fast mode (and its
    error): the switch exists for A/B, not for production."""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    from efficient_nerf_amd.create_data import RandStream, create_rand
    H = 64
    focal = O.focal_from_angle(H)
    sds = (_spiked(1), _spiked(2))

    class TopDownStream(RandStream):                     # poses from above only: the distribution the side probe did not see
        def rand_pose(self):
            from efficient_nerf_amd.frontend import pose_spherical
            theta = -180 + self.rs.rand() * 360
            self.rs.rand()
            return pose_spherical(theta, -88., 4)

    eng = NeRFEngine(H, H, focal).load_state_dicts(*sds)
    assert eng.choose_precision(*_rays(H, focal, O.pose_spherical(*HORIZONTAL, 4.)))[0] == 'fp16x1'
    tm, logs = {}, []
    d1 = str(tmp_path / 'watched')
    n = create_rand(eng, H, H, focal, 4, d1, i_save=2, split_size=512, stream=TopDownStream(), log=lambda *a: logs.append(' '.join(map(str, a))),
                    timings=tm)
    w = tm['watch']
    print(w, [l for l in logs if 'precision' in l])
    assert n == 2 * (2 * H * H // 512)
    assert w['fallbacks'] and w['fallbacks'][0]['pose'] == 1 and w['fallbacks'][0]['from'] == 'fp16x1' and w['precision'] != 'fp16x1'
    assert any('[precision] pose 1: fp16x1' in l for l in logs)
    final = w['precision']
    ref = NeRFEngine(H, H, focal, precision=PRECISIONS[final]).load_state_dicts(*sds)
    d2 = str(tmp_path / 'final_mode')
    create_rand(ref, H, H, focal, 4, d2, i_save=2, split_size=512, stream=TopDownStream(), log=lambda *a: None)
    for k in range(1, n + 1):
        assert open(os.path.join(d1, f'data_{k}.npy'), 'rb').read() == open(os.path.join(d2, f'data_{k}.npy'), 'rb').read(), k
    # and the unwatched run keeps what the probe chose
    eng.set_precision(PRECISIONS['fp16x1'])
    tm3 = {}
    create_rand(eng, H, H, focal, 2, str(tmp_path / 'unwatched'), i_save=2, split_size=512, stream=TopDownStream(), log=lambda *a: None,
                timings=tm3, watch=False)
    assert 'watch' not in tm3 and eng.precision_name == 'fp16x1'
    eng.close()
    ref.close()


def test_render_path_watch_falls_back_and_renders_the_frame_again(pkg, tmp_path):
    """frontend.render_path on a path that starts at the side and rises: the engine is in fp16x1 (probe on the first pose only, as
    round 4 did); with watch_every = 2 frame 2 (from above) misses, every later frame is rendered in the lower rung, and every
    watched frame is within the contract of the oracle"""
    from efficient_nerf_amd import NeRFEngine, frontend as fe
    H = 64
    focal = O.focal_from_angle(H)
    sds = (_spiked(1), _spiked(2))
    poses = [O.pose_spherical(30., ph, 4.) for ph in (-5., -8., -88., -86., -88.)]
    eng = NeRFEngine(H, H, focal).load_state_dicts(*sds)
    assert eng.choose_precision(*_rays(H, focal, poses[0]))[0] == 'fp16x1'
    st, logs = {}, []
    rgbs, _ = fe.render_path(poses, (H, H, focal), 'nerf', eng, log=lambda *a: logs.append(' '.join(map(str, a))), stats=st, watch_every=2)
    w = st['watch']
    print(w)
    assert w['fallbacks'] and w['fallbacks'][0]['frame'] == 2 and w['fallbacks'][0]['from'] == 'fp16x1' and w['precision'] != 'fp16x1'
    assert any('[precision] frame 2' in l for l in logs)
    idx = torch.arange(0, H * H, 13)
    for i in (2, 4):
        ro, rd = O.get_rays(H, H, focal, poses[i][:3, :4])
        want = O.render_rays(sds[0], sds[1], ro.reshape(-1, 3)[idx].float(), rd.reshape(-1, 3)[idx].float(), white_bkgd=True)['rgb_map']
        assert (rgbs[i].reshape(-1, 3)[idx].cpu() - want).abs().max().item() <= 1e-4, i
    eng.close()


@pytest.mark.parametrize('case', range(12))
def test_fp16x1_column_tilings_are_bitwise_equal(pkg, case):
    """The fp16x1 chain with 2, 3 and 4 sixteen-point column tiles per wave (128- / 192- / 256-point workgroup tiles; with GIVEN view
    directions, i.e. NDC renders, the four-tile request runs the three-tile build): the arithmetic per point does not depend on the
    tiling, so every output and extra is bit-identical -- ragged ray counts around the tile edges, random sample counts; and the
    tile count is a field of the context (it was a process-wide static)."""
    from efficient_nerf_amd import NeRFEngine, PREC_FP16X1, PREC_FP16X3
    from efficient_nerf_amd._lib import check, lib
    rng = np.random.RandomState(100 + case)
    S0 = int(rng.choice([3, 8, 16, 33, 64]))
    NI = min(int(rng.choice([1, 5, 32, 64, 128, 192])), 256 - S0)
    n = int([1, 63, 64, 65, 127, 129, 191, 193, 255, 257, 1000, 4099][case])
    ndc = case % 4 == 3
    seed = int(rng.randint(1, 1000))
    eng = NeRFEngine(8, 8, 10., near=0. if ndc else 2., far=1. if ndc else 6., N_samples=S0, N_importance=NI, white_bkgd=bool(case & 1),
                     precision=PREC_FP16X3, ndc=ndc).load_state_dicts(O.make_teacher_state(seed), O.make_teacher_state(seed + 1))
    other = NeRFEngine(8, 8, 10., N_samples=S0, N_importance=NI, white_bkgd=bool(case & 1), precision=PREC_FP16X1).load_state_dicts(O.make_teacher_state(seed), O.make_teacher_state(seed + 1))
    g = torch.Generator().manual_seed(case)
    ro = (torch.randn(n, 3, generator=g) * 0.3 + torch.tensor([0., 0., 4.])).cuda()
    rd = (torch.nn.functional.normalize(torch.randn(n, 3, generator=g) * 0.2 + torch.tensor([0., 0., -1.]), dim=-1) * (0.7 + 0.6 * torch.rand(n, 1, generator=g))).cuda()
    ref = eng.render_rays(ro, rd)['rgb_map'].clone()
    eng.set_precision(PREC_FP16X1)
    outs = {}
    for nc in (2, 3, 4):
        check(lib().nerf_debug_set_x1_col_tiles(eng._ctx, nc))
        outs[nc] = {k: v.clone() for k, v in eng.render_rays(ro, rd, extras=True).items()}
        # ... of THIS context only: another engine of the process still renders with its own (default) tiling, same bits
        if not ndc and nc == 2:
            o2 = other.render_rays(ro, rd)
            assert torch.equal(o2['rgb_map'], outs[2]['rgb_map'])
    # four column tiles run the one-statement build with the embedding in its stream (nerf_chain_emb_kernel, round 5) unless view
    # directions are given; with nerf_debug_set_x1_stream_embed(0) the round-4 build whose embedding is HIP code between two blocks
    check(lib().nerf_debug_set_x1_stream_embed(eng._ctx, 0))
    outs['hip'] = {k: v.clone() for k, v in eng.render_rays(ro, rd, extras=True).items()}
    check(lib().nerf_debug_set_x1_stream_embed(eng._ctx, 1))
    for k in outs[2]:
        assert torch.equal(outs[2][k], outs[4][k]) and torch.equal(outs[3][k], outs[4][k]) and torch.equal(outs['hip'][k], outs[4][k]), (k, n, S0, NI)
    assert (outs[4]['rgb_map'] - ref).abs().max().item() <= 1e-4
    with pytest.raises(Exception):
        check(lib().nerf_debug_set_x1_col_tiles(eng._ctx, 5))
    eng.close()
    other.close()
