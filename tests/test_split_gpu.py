"""GPU: R2L_PREC_FP16_SPLIT (round 5) -- the head layer and blocks [0, split) of the ResMLP in three fp16 passes, blocks [split, n_block)
on the generated bf6 kernel, two body launches handing the x image over (model/nerf_raybased.py:443-465, 539-544) -- and how
`--precision auto` uses it: for networks whose activations outgrow the bf6 / e4m3 rungs it measures the smallest split that stays within
its limit of three passes on every ray of a frame, and watches it."""
import os

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('mode', ['fp16_split', 'fp16_split8'])
def test_split_endpoints_and_middle_on_a_small_network(pkg, mode):
    """split = n_block is fp16x3_asm bit for bit (the same head launch, the same three-pass kernel over all blocks, the same fused
    tail); split = 0 is every block with bf6 terms behind the three-pass head: at least as close to the oracle's neighbourhood as
    fp16_fp8; every split inside the contract; a render of part of the rows equals the rows of the whole render (a ray's result does
    not depend on the launch it is in); the exponents of fp16_fp8 travel with the switch"""
    from efficient_nerf_amd import PREC_FP16_FP8, PREC_FP16X3_ASM, PRECISIONS, R2LEngine
    PREC_FP16_SPLIT = PRECISIONS[mode]
    H, nb = 48, 6
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=17, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(25., -35., 4.)
    ref = O.r2l_render(sd, H, H, focal, c2w)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    eng.calibrate_on(c2w=c2w)
    ex = eng.act_exponents()
    fp8 = eng.render(c2w).clone()
    eng.set_precision(PREC_FP16X3_ASM)
    x3 = eng.render(c2w).clone()
    eng.set_precision(PREC_FP16_SPLIT)
    assert eng.act_exponents() == ex                       # the calibrated exponents travel with the switch
    outs = {}
    for sp in (nb, 0, 3, 1, 5):
        eng.set_split_block(sp)
        outs[sp] = eng.render(c2w).clone()
        err = (outs[sp].cpu() - ref).abs().max().item()
        print(f'split {sp} of {nb}: L_inf vs CPU oracle {err:.2e}, vs fp16_fp8 {(outs[sp] - fp8).abs().max().item():.2e}, vs fp16x3_asm {(outs[sp] - x3).abs().max().item():.2e}')
        assert err <= 1e-4
    assert torch.equal(outs[nb], x3)
    d = {sp: (outs[sp] - x3).abs().max().item() for sp in outs}
    assert 0 < d[0] <= 3e-5 and d[5] <= d[0] + 2e-6       # mild i.i.d. weights: the bf6 terms cost little anywhere; fewer of them cost less
    eng.set_split_block(3)
    part = eng.render(c2w, rows=(7, 29))
    assert torch.equal(part, outs[3].view(H, H, 3)[7:29].reshape(-1, 3))
    with pytest.raises(Exception):
        eng.set_split_block(nb + 1)
    with pytest.raises(Exception):                         # its guarded launches see the blocks behind the split only
        eng.recalibrate()
    eng.close()


@pytest.mark.parametrize('mode', ['fp16_split', 'fp16_split8'])
def test_a_launch_over_part_of_the_blocks_converts_the_next_tile_with_its_own_first_exponent(pkg, mode):
    """the bf6 kernel converts the NEXT ray tile's x with the exponent its last block's aux names -- in the full stream block 0's input
    set, for a launch that starts at block `split` that block's (r2l_split_aux_kernel patches a copy of the aux blocks).  With
    exponents that rise with depth a wrong entry shows: a workgroup's second tile would differ from the same rays rendered as some
    workgroup's first tile.  450 ray tiles in one launch (two per workgroup) against two launches of 225 (one each): bit for bit"""
    from efficient_nerf_amd import PREC_FP16_FP8, PRECISIONS, R2LEngine
    PREC_FP16_SPLIT = PRECISIONS[mode]
    H, nb = 240, 6                                         # 57,600 rays = 450 ray tiles of 128
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=23, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(70., -25., 4.)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    ex = eng.calibrate_on(c2w=c2w)
    ex = ex[:6] + [e + 2 for e in ex[6:12]] + ex[12:]      # as in a trained network: the exponents rise with depth (i.i.d. weights: level)
    eng.set_act_exponents(ex)
    eng.set_precision(PREC_FP16_SPLIT)
    eng.set_split_block(nb)
    ref = eng.render(c2w).clone()
    eng.set_split_block(3)
    whole = eng.render(c2w).clone()
    halves = torch.cat([eng.render(c2w, rows=(0, H // 2)).clone(), eng.render(c2w, rows=(H // 2, H)).clone()], 0)
    d = (whole - ref).abs().max().item()
    print(f'split 3 of {nb}, exponents {ex}: {d:.2e} from three passes everywhere')
    assert torch.equal(whole, halves)
    assert d <= 1e-4
    eng.close()


def test_split_without_the_global_skip_and_in_a_graph(pkg):
    """networks without the global skip end in r2l_tail_kernel (three launches + the tail): the second body launch then continues
    the x image in place; and a warmed-up split render captures into a HIP graph and replays bit for bit"""
    from efficient_nerf_amd import PREC_FP16_SPLIT, R2LEngine
    H, nb = 32, 4
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=3, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(-50., -20., 4.)
    eng = R2LEngine(H, H, focal, n_block=nb, use_residual=False, precision=PREC_FP16_SPLIT).load_state_dict(sd)
    eng.set_split_block(2)
    got = eng.render(c2w).clone()
    pts = O.sample_test(O.camera_dirs(H, H, focal), O.sampler_z_vals(16, 2., 6.), c2w[:3, :4])
    ref = O.r2l_forward(sd, O.positional_embed(pts, 10), use_residual=False)
    assert (got.cpu() - ref).abs().max().item() <= 1e-4
    eng.close()
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_SPLIT).load_state_dict(sd)
    assert eng.split_block == nb // 2                      # the default, on both sides of the C-ABI
    half = eng.render(c2w).clone()
    eng.set_split_block(2)
    assert torch.equal(eng.render(c2w), half)
    eng.set_guard_period(0)
    first = eng.render(c2w).clone()
    out = torch.empty_like(first)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.render(c2w, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            eng.render(c2w, out=out)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, first)
    eng.close()


@pytest.mark.parametrize('mode', ['fp16_split', 'fp16_split8'])
def test_random_networks_frames_and_splits(pkg, mode):
    """randomised: depth, ragged frame shapes (ray counts that are no multiple of the 128-ray tile), pose, global skip on / off, every
    split of the depth: split = n_block reproduces fp16x3_asm bit for bit, split = 0 differs from fp16_fp8 by the head launch only,
    everything in between stays as close to three passes as fp16_fp8 does on these i.i.d. weights"""
    from efficient_nerf_amd import PREC_FP16_FP8, PREC_FP16X3_ASM, PRECISIONS, R2LEngine
    PREC_FP16_SPLIT = PRECISIONS[mode]
    rng = np.random.default_rng(5)
    worst = 0.0
    for it in range(8):
        H, W = int(rng.integers(3, 70)), int(rng.integers(3, 70))
        nb = int(rng.integers(1, 30))
        res = bool(rng.integers(0, 2))
        sd = O.make_r2l_state(seed=int(rng.integers(1 << 30)), netdepth=2 + 2 * nb, body_gain=float(rng.choice([0.7, 1.0])))
        c2w = O.pose_spherical(float(rng.uniform(-180, 180)), float(rng.uniform(-80, -5)), float(rng.uniform(3, 5)))
        focal = O.focal_from_angle(W)
        eng = R2LEngine(H, W, focal, n_block=nb, use_residual=res, precision=PREC_FP16_FP8).load_state_dict(sd)
        eng.calibrate_on(c2w=c2w)
        fp8 = eng.render(c2w).clone()
        eng.set_precision(PREC_FP16X3_ASM)
        x3 = eng.render(c2w).clone()
        eng.set_precision(PREC_FP16_SPLIT)
        limit = max(6e-5, 2.0 * (fp8 - x3).abs().max().item())
        for sp in sorted({0, nb, int(rng.integers(0, nb + 1)), int(rng.integers(0, nb + 1))}):
            eng.set_split_block(sp)
            got = eng.render(c2w)
            d = (got - x3).abs().max().item()
            assert torch.isfinite(got).all() and d <= limit, (it, H, W, nb, res, sp, d, limit)
            if sp == nb:
                assert torch.equal(got, x3), (it, H, W, nb, res)
            worst = max(worst, d)
        eng.close()
    print(f'8 random networks x up to 4 splits: worst {worst:.2e} from three passes everywhere')


def test_auto_measures_the_split_on_the_trained_like_student(pkg):
    """the committed trained-like student (max|a| 126: beyond the whole-network bf6 and e4m3 rungs): per format behind the split (bf6
    terms, e4m3 terms) `auto` bisects for the fewest leading blocks in three passes whose frame stays within its limit of three passes
    everywhere, on every ray of a frame, and takes the cheaper of the two; the frames it then renders are inside the 1e-4 contract of
    the CPU oracle; the watch's spot check agrees; the step-down ends in fp16x3_asm; a limit of zero ends there at once, as before"""
    from efficient_nerf_amd import PREC_FP16_SPLIT, PREC_FP16_SPLIT8, PREC_NAMES, R2LEngine
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trained_like', 'student_w256d88.npz'))
    ssd = {k: torch.from_numpy(z[k]) for k in z.files}
    H = 400
    focal = O.focal_from_angle(H)
    test = O.novel_poses(200)
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(ssd)
    rung, top = eng.choose_precision(c2w=test[0][:3, :4])
    print(f'trained-like student: max|a| {eng.stream_max:.1f} -> {rung}, split {eng.split_block}; measured {eng.auto_split}')
    sp = eng.split_block
    assert rung in ('fp16_split', 'fp16_split8') and 0 <= sp <= 37 and eng.auto_split[rung][sp] <= eng.AUTO_SPLIT_MAX_DIFF
    assert sp == 0 or any(k < sp and v > eng.AUTO_SPLIT_MAX_DIFF for k, v in eng.auto_split[rung].items())     # a smaller one was tried, and failed
    # the other format was measured too (bf6 first), and what was taken is the cheaper by the measured block times
    assert set(eng.auto_split) == {'fp16_split', 'fp16_split8'}
    ok6 = [k for k, v in eng.auto_split['fp16_split'].items() if v <= eng.AUTO_SPLIT_MAX_DIFF]
    assert ok6 and eng.split_cost(eng.precision, sp) <= eng.split_cost(PREC_FP16_SPLIT, min(ok6))
    from efficient_nerf_amd import get_rays
    for pi in (0, 67, 133):
        got, again = eng.render_checked(lambda: eng.render(test[pi][:3, :4]))
        assert PREC_NAMES[eng.precision] == rung
        g = got.cpu().view(H, H, 3)[::8].reshape(-1, 3)
        want = O.r2l_render(ssd, H, H, focal, test[pi][:3, :4], rows=(0, H, 8), chunk=16384)
        err = (g - want).abs().max().item()
        ro, rd = get_rays(H, H, focal, test[pi][:3, :4], device='cuda')
        ok, d = eng.spot_check_split(ro, rd)
        print(f'pose {pi}: L_inf vs CPU oracle on {g.shape[0]} rays {err:.2e}; spot check {d:.2e}')
        assert err <= 1e-4 and ok
    # the step-down of the watch: half of the low-precision part to three passes, until too little would be saved
    assert eng.step_down_split() == rung and eng.split_block == sp + (43 - sp + 1) // 2
    while eng.step_down_split() == rung:
        assert eng.split_block < 43
    assert PREC_NAMES[eng.precision] == 'fp16x3_asm' and eng.split_block is None
    eng.close()
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(ssd)
    eng.AUTO_SPLIT_MAX_DIFF = 0.0
    assert eng.choose_precision(c2w=test[0][:3, :4])[0] == 'fp16x3_asm' and eng.split_block is None
    eng.close()


def test_auto_verifies_the_rung_the_activation_limits_name(pkg):
    """a ReLU ResMLP is positively homogeneous: head and biases x 1/32, tail weight x 32 is the same function with activations 32 times
    smaller.  The trained-like student scaled that way has max|a| = 3.9 -- inside fp16_fp8's activation limit (8), which was derived from
    i.i.d. weight families -- and the same 1.3e-4 error in fp16_fp8, because what makes the error is what the network does with it, not
    the size of its stream.  `auto` renders the rung the limits name against three passes on every ray of the probe frame, sees it, and
    takes the measured rungs; the synthetic W256D88 weights pass the same check and keep fp16_fp8"""
    from efficient_nerf_amd import PREC_FP16_FP8, PREC_NAMES, R2LEngine
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trained_like', 'student_w256d88.npz'))
    ssd = {k: torch.from_numpy(z[k]).clone() for k in z.files}
    a = 1. / 32.
    for k in ssd:
        if k.startswith('head.') or (k.startswith('body.') and k.endswith('bias')):
            ssd[k] = ssd[k] * a
        elif k == 'tail.0.weight':
            ssd[k] = ssd[k] / a
    H = 400
    focal = O.focal_from_angle(H)
    test = O.novel_poses(200)
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(ssd)
    rung, top = eng.choose_precision(c2w=test[0][:3, :4])
    print(f'scaled trained-like student: max|a| {eng.stream_max:.2f} (exponent {top}); fp16_fp8 is {eng.auto_verify:.2e} from three passes -> {rung}, split {eng.split_block}')
    assert eng.stream_max <= eng.AUTO_MAX_ABS and eng.auto_verify > eng.AUTO_VERIFY_MAX_DIFF
    assert rung in ('fp16_split', 'fp16_split8', 'fp16x3_asm') and PREC_NAMES[eng.precision] == rung
    got = eng.render(test[67][:3, :4]).cpu().view(H, H, 3)[::8].reshape(-1, 3)
    want = O.r2l_render(ssd, H, H, focal, test[67][:3, :4], rows=(0, H, 8), chunk=16384)
    assert (got - want).abs().max().item() <= 1e-4
    eng.set_precision(PREC_FP16_FP8)         # what the limits alone would have rendered it with
    bad = (eng.render(test[67][:3, :4]).cpu().view(H, H, 3)[::8].reshape(-1, 3) - want).abs().max().item()
    print(f'pose 67 against the CPU oracle: {rung} {(got - want).abs().max().item():.2e}, fp16_fp8 {bad:.2e}')
    eng.close()
    sd = O.make_r2l_state(seed=0)
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(sd)
    rung, top = eng.choose_precision(c2w=test[0][:3, :4])
    print(f'synthetic weights: max|a| {eng.stream_max:.2f}; fp16_fp8 is {eng.auto_verify:.2e} from three passes -> {rung}')
    assert rung == 'fp16_fp8' and eng.auto_verify <= eng.AUTO_VERIFY_MAX_DIFF and eng.auto_split is None
    eng.close()


def test_whole_network_rung_is_watched_and_falls_back_to_the_measured_rungs(pkg):
    """fp16_fp8 chosen by `auto` used to be watched for activation RANGE only.  The scaled trained-like student
    (max|a| 3.9: inside fp16_fp8's activation limit, 1.3e-4 off in it on whole frames) is probed on the rays of a frame where fp16_fp8 happens
    to be good (<= 2e-5 from three passes): `auto` verifies inside its limit there and keeps fp16_fp8 -- what a probe pose that does not see
    the hard rays would do.  The rgb watch (spot_check_rgb: a sample of the rendered rays against a second context in three passes) must
    see the other rays of the frame miss, step down to the measured split rungs, and what is rendered afterwards must be inside the 1e-4
    contract of the CPU oracle; render_path must do all of that by itself on its first batch.  Synthetic weights: the watch passes."""
    from efficient_nerf_amd import PREC_FP16_FP8, PREC_FP16X3_ASM, PREC_NAMES, R2LEngine, get_rays
    from efficient_nerf_amd import frontend as fe
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'trained_like', 'student_w256d88.npz'))
    ssd = {k: torch.from_numpy(z[k]).clone() for k in z.files}
    a = 1. / 32.
    for k in ssd:
        if k.startswith('head.') or (k.startswith('body.') and k.endswith('bias')):
            ssd[k] = ssd[k] * a
        elif k == 'tail.0.weight':
            ssd[k] = ssd[k] / a
    H = 400
    focal = O.focal_from_angle(H)
    test = O.novel_poses(200)
    pose = test[0][:3, :4]
    ro, rd = (t.reshape(-1, 3) for t in get_rays(H, H, focal, pose, device='cuda'))
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(ssd)
    eng.set_precision(PREC_FP16_FP8)
    eng.calibrate_on(c2w=pose)
    got8 = eng.render(pose).clone()
    eng.set_precision(PREC_FP16X3_ASM)
    d = (got8 - eng.render(pose)).abs().max(-1)[0]
    easy = torch.nonzero(d <= 2e-5).flatten()[:65536]
    print(f'scaled trained-like student: fp16_fp8 is {d.max().item():.2e} from three passes on the frame; {int((d <= 2e-5).sum())} rays within 2e-5')
    assert d.max().item() > eng.WHOLE_WATCH_MAX_DIFF and easy.numel() >= 16384
    rung, top = eng.choose_precision(rays=(ro[easy].contiguous(), rd[easy].contiguous()))
    print(f'probed on {easy.numel()} easy rays: verify {eng.auto_verify:.2e} -> {rung}')
    assert rung == 'fp16_fp8' and eng.auto_verify <= eng.AUTO_VERIFY_MAX_DIFF and eng.watched_mode() == 'whole'
    # the engine-level watch: the frame's rays miss, the step-down lands on a measured rung that passes
    ok, dd = eng.spot_check_rgb(ro, rd)
    print(f'spot_check_rgb on the whole frame: ok {ok}, {dd:.2e} (limit {eng.WHOLE_WATCH_MAX_DIFF:g})')
    assert not ok and dd > eng.WHOLE_WATCH_MAX_DIFF
    now = eng.step_down_whole(ro, rd)
    print(f'step_down_whole -> {now}, split {eng.split_block}; measured {eng.auto_split}')
    assert now in ('fp16_split', 'fp16_split8', 'fp16x3_asm') and PREC_NAMES[eng.precision] == now
    ok, dd = eng.spot_check_rgb(ro, rd)
    assert ok, dd
    got = eng.render(test[67][:3, :4]).cpu().view(H, H, 3)[::8].reshape(-1, 3)
    want = O.r2l_render(ssd, H, H, focal, test[67][:3, :4], rows=(0, H, 8), chunk=16384)
    print(f'pose 67 in {now}: {(got - want).abs().max().item():.2e} from the CPU oracle')
    assert (got - want).abs().max().item() <= 1e-4
    eng.close()
    # render_path by itself: same probe, then three frames; the first batch's watch trips and the frames it returns are the re-rendered ones
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(ssd)
    assert eng.choose_precision(rays=(ro[easy].contiguous(), rd[easy].contiguous()))[0] == 'fp16_fp8'
    lines, stats = [], {}
    rgbs, _ = fe.render_path([test[i] for i in (0, 67, 133)], (H, H, focal), 'R2L', eng, log=lines.append, stats=stats)
    w = stats['split_watch']
    print('\n'.join(ln for ln in lines if 'precision' in ln))
    print({k: v for k, v in w.items()})
    assert len(w['fallbacks']) >= 1 and w['fallbacks'][0]['from'] == 'fp16_fp8' and w['fallbacks'][0]['frame'] == 0
    assert eng.precision_name in ('fp16_split', 'fp16_split8', 'fp16x3_asm') and stats['rerenders'] >= 1
    for j, pi in enumerate((0, 67, 133)):
        want = O.r2l_render(ssd, H, H, focal, test[pi][:3, :4], rows=(0, H, 8), chunk=16384)
        err = (rgbs[j].cpu()[::8].reshape(-1, 3) - want).abs().max().item()
        print(f'render_path frame {j} (pose {pi}) after the fallback: {err:.2e} from the CPU oracle')
        assert err <= 1e-4
    eng.close()
    # synthetic weights: fp16_fp8 stays, the watch runs and passes
    sd = O.make_r2l_state(seed=0)
    eng = R2LEngine(H, H, focal, 2., 6., n_block=43, use_residual=True).load_state_dict(sd)
    assert eng.choose_precision(c2w=pose)[0] == 'fp16_fp8' and eng.watched_mode() == 'whole'
    stats = {}
    fe.render_path([test[i] for i in range(0, 200, 10)], (H, H, focal), 'R2L', eng, log=lambda s: None, stats=stats)
    w = stats['split_watch']
    print(f'synthetic weights, 20 frames: {w}')
    assert w['checks'] == 2 and not w['fallbacks'] and w['worst'] <= eng.WHOLE_WATCH_MAX_DIFF and eng.precision_name == 'fp16_fp8'
    eng.close()
