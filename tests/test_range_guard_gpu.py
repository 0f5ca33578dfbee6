"""Range tracking of R2L_PREC_FP16_FP8 (include/r2l_hip.h: r2l_range_status, r2l_set_guard_period, r2l_recalibrate).

The bf6 correction terms are scaled per operand set by exponents measured once; the contract (rgb L_inf <= 1e-4 vs the
reference, BASELINE.json north_star) must hold for EVERY ray rendered afterwards, at any pose (the reference renders any
pose with one model call: main.py:300-309).  So the library measures instead of assuming: the head launch tracks h0 of
every ray, the range-guard build of the body kernel tracks all 2 n_block operand sets of every ray of the launches it
runs for, and `--precision auto` re-checks after every frame."""
import math

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def oracle_set_maxima(sd, emb, n_block):
    """float64 maxima of the operand sets the body kernel converts to bf6: IN_b = |x~_b| with x~_b = x_b - sum_{j<b} b2_j
    (the layer-2 biases are folded on the host, csrc/r2l_capi.hip pack_body_v3), H_b = relu(W1 x_b + b1)"""
    g = lambda k: sd[k].double()
    x = torch.relu(emb.double() @ g('head.0.weight').T + g('head.0.bias'))
    h0max = x.max().item()
    bsum = torch.zeros(256, dtype=torch.float64)
    out = []
    for b in range(n_block):
        out.append((x - bsum).abs().max().item())
        h = torch.relu(x @ g(f'body.{b}.body.0.weight').T + g(f'body.{b}.body.0.bias'))
        out.append(h.max().item())
        x = x + h @ g(f'body.{b}.body.2.weight').T + g(f'body.{b}.body.2.bias')
        bsum = bsum + g(f'body.{b}.body.2.bias')
    return h0max, out


def exponent_of(m):
    """r2l_calib_finalize_kernel: the smallest E with max * 16 / 2^E <= 16"""
    fr, e = math.frexp(m)
    return e - 1 if fr == 0.5 else e


def frame_embedding(H, W, focal, c2w, idx=None):
    pts = O.sample_test(O.camera_dirs(H, W, focal), O.sampler_z_vals(16, 2., 6.), torch.as_tensor(c2w)[:3, :4])
    return O.positional_embed(pts if idx is None else pts[idx])


def test_guarded_launch_is_bit_identical_and_measures_every_set(pkg):
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 64, 5
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=3, netdepth=2 + 2 * nb)
    for k in sd:                                  # growing ranges: every set gets its own exponent
        if k.startswith('body.') and k.endswith('weight'):
            sd[k] = sd[k] * 1.3
    c2w = O.pose_spherical(70., -25., 4.)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    ex = eng.calibrate_on(c2w=c2w)                # exponents from every ray of the frame
    h0max, sets = oracle_set_maxima(sd, frame_embedding(H, H, focal, c2w), nb)
    for j, m in enumerate(sets):                  # the kernel's maxima carry its own 1e-3-grade rounding: only a maximum
        want = exponent_of(m)                     # within that of a power of two may land on the other side
        assert ex[j] == want or abs(m / 2.0 ** round(math.log2(m)) - 1) < 2e-3, (j, ex[j], want, m)
    eng.set_guard_period(0)
    plain = eng.render(c2w).cpu()
    st = eng.range_status(reset=True)
    assert st['launches'] == 1 and st['guarded_launches'] == 0 and st['worst_set'] == -1
    assert abs(st['h0_max'] - h0max) <= 2e-3 * h0max          # the head tracks every ray in every launch
    eng.set_guard_period(1)
    guarded = eng.render(c2w).cpu()
    st = eng.range_status(reset=True)
    assert torch.equal(plain, guarded)
    assert st['launches'] == 1 and st['guarded_launches'] == 1
    fills = [m * 16 / 2.0 ** ex[j] / 28 for j, m in enumerate(sets)]
    assert st['worst_set'] == int(np.argmax(fills)) or abs(st['worst_fill'] - max(fills)) < 2e-3
    assert abs(st['worst_fill'] - max(fills)) <= 2e-3 and 0.28 < st['worst_fill'] <= 16 / 28 + 1e-3
    assert not st['saturated'] and not st['beyond_calibration']
    # every k-th launch: the first after the period is set, then launches k, 2k, ...
    eng.set_guard_period(3)
    for _ in range(7):
        eng.render(c2w)
    st = eng.range_status(reset=True)
    assert (st['launches'], st['guarded_launches']) == (7, 3)
    assert eng.check_ranges() is None             # nothing left the scales
    eng.close()


def test_edge_rays_beyond_the_centre_probe_are_caught(pkg):
    """A network whose activations grow towards the image border: eight head units compute relu(12 (d . up - 0.1)) from the
    identity columns of head.0.weight (the point coordinates, model/nerf_raybased.py:206; the difference of a ray's last
    and first point is 4 d) -- zero for the rows around the image centre, up to 3 at the top rows -- and the first
    ResMLP block multiplies them by 60.  A probe through the centre, which is how round 2 chose the exponents, reads
    too small a range.  Rendering the whole frame with those
    exponents must raise the flags; check_ranges must raise the exponents and ask for the frame again; calibrate_on
    must see the border rays in the first place."""
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 200, 43
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=0)
    c2w = torch.as_tensor(O.pose_spherical(0., -30., 4.))[:3, :4].float()
    up, gain = c2w[:, 1], 12.0
    Wh, bh = sd['head.0.weight'].clone(), sd['head.0.bias'].clone()
    W1, b1 = sd['body.0.body.0.weight'].clone(), sd['body.0.body.0.bias'].clone()
    for r in range(8):
        Wh[r] = 0
        for c in range(3):
            Wh[r, (3 * 15 + c) * 21 + 20] = gain / 4 * up[c]
            Wh[r, c * 21 + 20] = -gain / 4 * up[c]
        bh[r] = -gain * 0.1
        W1[r], b1[r] = 0, 0
        W1[r, r] = 60.0
    sd['head.0.weight'], sd['head.0.bias'], sd['body.0.body.0.weight'], sd['body.0.body.0.bias'] = Wh, bh, W1, b1
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    eng.set_guard_period(1)
    band = (H // 2 - 3, H // 2 + 3)             # 1,200 rays through the centre: the library's own sample only
    eng.render(c2w, rows=band)
    probe = eng.act_exponents()
    eng.range_status(reset=True)
    eng.render(c2w)                              # the frame under the probe's exponents
    st = eng.range_status()
    assert st['beyond_calibration'] and st['saturated'] and st['worst_fill'] > 1 and st['worst_set'] >= 1, st
    logs = []
    assert eng.check_ranges(log=logs.append) == 'fp16_fp8' and logs and 'values were clamped' in logs[0], logs
    raised = eng.act_exponents()
    assert all(b >= a for a, b in zip(probe, raised)) and raised[1] >= probe[1] + 3, (probe, raised)
    eng.render(c2w)
    assert eng.check_ranges() is None            # the second render of the frame is inside the new scales
    st = eng.range_status()
    assert not st['saturated'] and max(st['h0_fill'], st['worst_fill']) <= 16 / 28 + 2e-3, st
    full = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd).calibrate_on(c2w=c2w)
    assert full == raised, (full, raised)
    # what `--precision auto` makes of it: the border rays put the exponents past fp16_fp8's limit
    name, top = eng.choose_precision(c2w=c2w)
    assert name in ('fp16_split', 'fp16_split8', 'fp16x3_asm') and top == max(full) > eng.AUTO_MAX_EXP      # behind the whole-network rungs: the split, measured
    ref = O.r2l_render(sd, H, H, focal, c2w, rows=(0, 8))
    assert (eng.render(c2w, rows=(0, 8)).cpu() - ref).abs().max().item() <= 1e-4
    eng.close()


@pytest.mark.parametrize('gain,want', [(1.08, 'fp16_e4m3'), (1.3, 'fp16x3_asm')])
def test_auto_falls_back_when_a_later_pose_leaves_the_range(pkg, gain, want):
    """`--precision auto` = choose_precision + check_ranges after every frame: exponents forced low (as if the first frame
    had been a tame one) -> the next frame trips the watch, the context moves down the ladder (fp16_fp8 -> fp16_e4m3 at
    exponent 4 -> fp16x3_asm above), and the re-rendered frame meets the contract."""
    from efficient_nerf_amd import PRECISIONS, R2LEngine
    H = 96
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=0)
    for k in sd:
        if k.startswith('body.') and k.endswith('weight'):
            sd[k] = sd[k] * gain                  # largest |activation| ~ 9 (gain 1.08: the e4m3 rung) / 5-6 (1.3): beyond fp16_fp8's limit
    c2w = O.pose_spherical(40., -30., 4.)
    eng = R2LEngine(H, H, focal).load_state_dict(sd)
    name, top = eng.choose_precision(c2w=c2w, max_exp=8)      # limit lifted: stays in fp16_fp8 ...
    assert name == 'fp16_fp8' and top > eng.AUTO_MAX_EXP
    eng._auto = (None,)                                        # ... then the real ladder, with exponents two too small
    eng.set_act_exponents([e - 2 for e in eng.act_exponents()])
    eng.set_guard_period(1)
    eng.range_status(reset=True)
    eng.render(c2w)
    logs = []
    assert eng.check_ranges(log=logs.append) == want, logs
    assert eng.precision == PRECISIONS[want] and want in logs[0]
    ref = O.r2l_render(sd, H, H, focal, c2w)
    assert (eng.render(c2w).cpu() - ref).abs().max().item() <= 1e-4
    assert eng.check_ranges() is None            # inside the scales now (fp16x3_asm has none to watch)
    # straight through choose_precision: the same rung
    eng2 = R2LEngine(H, H, focal).load_state_dict(sd)
    assert eng2.choose_precision(c2w=c2w)[0] == want
    assert (eng2.render(c2w).cpu() - ref).abs().max().item() <= 1e-4
    eng.close()
    eng2.close()


def test_calibrate_on_one_pose_holds_the_contract_on_the_whole_test_path(pkg):
    """Exponents from every ray of ONE 800x800 frame (test pose 0), then eight poses spread over the 200-view test path
    (load_blender.py:327-333), each checked against the CPU oracle on 2,500 strided rays and watched by the range guard."""
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H = 800
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=0)
    poses = O.novel_poses(200)[:, :3, :4]
    eng = R2LEngine(H, H, focal, precision=PREC_FP16_FP8).load_state_dict(sd)
    ex = eng.calibrate_on(c2w=poses[0])
    assert max(ex) <= eng.AUTO_MAX_EXP
    eng.set_guard_period(1)
    idx = torch.arange(0, H * H, H * H // 2500)[:2500]
    worst, fill = 0.0, 0.0
    for pi in range(0, 200, 25):
        rgb = eng.render(poses[pi]).cpu()
        st = eng.range_status(reset=True)
        assert not st['saturated'] and max(st['worst_fill'], st['h0_fill']) < eng.FILL_LIMIT, (pi, st)
        fill = max(fill, st['worst_fill'], st['h0_fill'])
        ref = O.r2l_forward(sd, frame_embedding(H, H, focal, poses[pi], idx))
        worst = max(worst, (rgb[idx] - ref).abs().max().item())
    print('8 poses of the test path under pose-0 exponents: L_inf %.2e, largest fill %.3f (calibration aims at 0.571)' % (worst, fill))
    assert worst <= 1e-4, worst
    eng.close()


def test_unguarded_frame_that_clamps_is_measured_then_raised(pkg):
    """at the default guard period 7 of 8 launches track only the head output.  A frame whose h0 was clamped on
    such a launch must not be emitted: check_ranges asks for a range-guarded render of the frame ('measure'), raises the
    exponents from what that render saw, asks for the frame once more (values had been clamped) and only then lets it
    stand.  `render_checked` is the loop frontend.render_path runs."""
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 96, 6
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=5, netdepth=2 + 2 * nb)
    c2w = O.pose_spherical(20., -40., 4.)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    good = eng.calibrate_on(c2w=c2w)
    assert eng._guard_period == 8
    eng.render(c2w)                                   # launch 0 of the guard's phase: guarded; the next seven are not
    eng.set_act_exponents([e - 2 for e in good])      # as if the calibration frame had been a tame one
    eng.range_status(reset=True)
    logs = []
    seq = []

    def check():
        r = eng.check_ranges(log=logs.append)
        seq.append(r)
        return r

    rgb, again = eng.render_checked(lambda: eng.render(c2w), check=check)
    assert seq[0] == 'measure' and seq[1] == 'fp16_fp8' and seq[-1] is None and again == len(seq) - 1 <= 3, (seq, logs)
    assert 'only the head output' in logs[0] and 'values were clamped' in logs[0] and 'exponents raised' in logs[1], logs
    assert eng.range_status()['guarded_launches'] == 1      # the frame that stands was watched in every operand set
    # the maxima came from a pass with clamped correction terms: within one of the clean calibration, never clamping
    assert all(abs(a - b) <= 1 for a, b in zip(eng.act_exponents(), good)), (eng.act_exponents(), good)
    ref = O.r2l_render(sd, H, H, focal, c2w)
    assert (rgb.cpu() - ref).abs().max().item() <= 1e-4
    eng.close()


def test_e4m3_limit_is_relative_to_the_calibrated_range(pkg):
    """e4m3 clamps at 448, 28 x the value the calibration aims at, so a limit of 0.9 x the format's top would let
    activations grow 25-fold -- far past exponent 4, the last one the ladder admits e4m3 at -- before anything acted.  The
    limit is the bf6 one in units of the scale (25.2 against the calibrated 16): exponents two too small (values up to 64,
    nothing clamped) must trip it, and `auto` must move on to fp16x3_asm when the raised exponents pass 4."""
    from efficient_nerf_amd import PREC_FP16_E4M3, PRECISIONS, R2LEngine
    H = 96
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=0)
    for k in sd:
        if k.startswith('body.') and k.endswith('weight'):
            sd[k] = sd[k] * 1.3                   # exponents 5-6
    c2w = O.pose_spherical(40., -30., 4.)
    eng = R2LEngine(H, H, focal).load_state_dict(sd)
    eng.set_precision(PREC_FP16_E4M3)
    true_ex = eng.calibrate_on(c2w=c2w)
    assert max(true_ex) > eng.AUTO_MAX_EXP_E4M3
    eng._auto = (None,)                           # the real ladder, with exponents two too small
    eng.set_act_exponents([e - 2 for e in true_ex])
    eng.set_guard_period(1)
    eng.range_status(reset=True)
    eng.render(c2w)
    st = eng.range_status()
    assert not st['saturated'] and 25.2 < st['worst_fill'] * st['format_top'] <= 64 * 1.01, st
    assert abs(eng.fill_limit(st['format_top']) - 25.2 / 448) < 1e-9
    logs = []
    assert eng.check_ranges(log=logs.append) == 'fp16x3_asm', logs
    assert eng.precision == PRECISIONS['fp16x3_asm']
    ref = O.r2l_render(sd, H, H, focal, c2w)
    assert (eng.render(c2w).cpu() - ref).abs().max().item() <= 1e-4
    eng.close()


def test_range_status_is_one_copy(pkg):
    """range word, maxima and exponents live in one allocation; what r2l_get_range_status and
    r2l_get_act_exponents report must agree with r2l_set_act_exponents and with a device-side calibration"""
    from efficient_nerf_amd import PREC_FP16_FP8, R2LEngine
    H, nb = 48, 5
    focal = O.focal_from_angle(H)
    sd = O.make_r2l_state(seed=8, netdepth=2 + 2 * nb)
    eng = R2LEngine(H, H, focal, n_block=nb, precision=PREC_FP16_FP8).load_state_dict(sd)
    want = [3, 1, 2, 0, 4, -1, 2, 2, 1, 0, 3]
    eng.set_act_exponents(want)
    assert eng.act_exponents() == want
    c2w = O.pose_spherical(0., -30., 4.)
    a = eng.render(c2w).cpu()
    assert eng.act_exponents() == want                      # a render under fixed exponents leaves them alone
    eng.set_act_exponents(None)                             # measure again: the calibration kernels write both places
    eng.render(c2w)
    ex = eng.act_exponents()
    assert ex != want and ex[-1] == ex[0]
    eng.set_act_exponents(ex)                               # spread kernel writes what the calibration wrote
    b = eng.render(c2w).cpu()
    eng.set_act_exponents(None)
    eng.render(c2w)
    assert eng.act_exponents() == ex and torch.equal(eng.render(c2w).cpu(), b)
    assert (a - b).abs().max().item() < 1e-3
    eng.close()
