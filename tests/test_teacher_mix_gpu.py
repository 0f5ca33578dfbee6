"""GPU: R2L_PREC_FP16_MIX (round 6) -- the teacher's fine network on the bf6 layer chain with its first two trunk layers in three fp16
passes (csrc/gen/nerf_gen.py NERF_GEN_FMT=mix, emulated on the CPU by tests/test_nerf_genx_cpu.py), the coarse network in fp16x3_asm.
The rung `--precision auto` gives TRAINED teachers instead of three passes everywhere: the reference's teacher is
always a trained network (model/nerf_raybased.py:337-401 through main.py:624-756; create_data: utils/create_data.py:812-872)."""
import os
import time

import numpy as np
import pytest
import torch

from oracle import r2l_oracle as O

pytestmark = pytest.mark.gpu


def test_mix_raw_against_fp16x3_on_synthetic_weights(pkg):
    """run_network in the mixed chain (both networks) against the compiler-scheduled fp16x3: raw within the bf6 chain's tolerance"""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    H = 40
    focal = O.focal_from_angle(H)
    sds = (O.make_teacher_state(1), O.make_teacher_state(2))
    eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3']).load_state_dicts(*sds)
    ro, rd = O.get_rays(H, H, focal, O.pose_spherical(30., -30., 4.)[:3, :4])
    ro, rd = ro.reshape(-1, 3).float().cuda(), rd.reshape(-1, 3).float().cuda()
    z = eng.z_coarse.cuda()
    ref = [eng.run_network(w, ro, rd, z).clone() for w in (0, 1)]
    eng.set_precision_pair(PRECISIONS['fp16_mix'], PRECISIONS['fp16_mix'])
    for w in (0, 1):
        d = (eng.run_network(w, ro, rd, z) - ref[w]).abs().max().item()
        print(f'network {w}: raw of the mixed chain within {d:.2e} of fp16x3')
        assert d <= 2e-4
    # a render with the pair the rung uses: coarse fp16x3_asm, fine fp16_mix
    eng.set_precision(PRECISIONS['fp16_mix'])
    assert eng.precision_name == 'fp16_mix' and eng.precision_coarse == PRECISIONS['fp16x3_asm']
    got = eng.render(O.pose_spherical(30., -30., 4.))
    eng.set_precision(PRECISIONS['fp16x3'])
    want = eng.render(O.pose_spherical(30., -30., 4.))
    for k in ('rgb_map', 'acc_map'):
        assert (got[k] - want[k]).abs().max().item() <= 2e-5, k
    eng.close()
    e2 = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16_mix']).load_state_dicts(*sds)       # constructed in the rung: the same pair
    assert e2.precision_name == 'fp16_mix' and e2.precision_coarse == PRECISIONS['fp16x3_asm']
    assert torch.equal(e2.render(O.pose_spherical(30., -30., 4.))['rgb_map'], got['rgb_map'])
    e2.close()


def test_trained_like_teacher_gets_the_mixed_rung_and_it_holds_on_whole_frames(pkg):
    """the committed trained-like teacher: `auto` (create_data's probes) ends on fp16_mix; against three passes for both networks on the
    same generated chain -- the same coarse pass, so z_samples are bitwise the same and the maps compare directly -- every ray of three
    whole 400 x 400 frames is within 5e-5 on rgb; the frame is faster than three passes everywhere; the watch passes; one rung down is
    fp16x3_asm"""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    from efficient_nerf_amd import create_data as CD
    from oracle import whole_frame as WF
    sds = WF.load_teacher()
    H = WF.H
    eng = NeRFEngine(H, H, WF.focal(), precision=PRECISIONS['fp16x3']).load_state_dicts(*sds)
    name = CD.choose_precision_for_rand(eng, H, H, WF.focal())
    print('auto:', name, eng.auto_diffs)
    assert name == 'fp16_mix' and eng.precision_name == 'fp16_mix'
    assert eng.auto_diffs['fp16_mix'] <= eng.AUTO_MAX_DIFF_MIX and eng.auto_diffs['fp16x3_asm'] <= eng.AUTO_MAX_DIFF_X3ASM
    assert eng.auto_diffs['fp16_fp8'] > eng.AUTO_MAX_DIFF
    times = {}
    outs = {}
    for mode in ('fp16x3_asm', 'fp16_mix'):
        eng.set_precision(PRECISIONS[mode])
        eng.render(WF.pose(0))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs[mode] = [{k: v.clone() for k, v in eng.render(WF.pose(pi), extras=True).items()} for pi in range(3)]
        torch.cuda.synchronize()
        times[mode] = (time.perf_counter() - t0) / 3
    worst = 0.
    for pi in range(3):
        a, b = outs['fp16_mix'][pi], outs['fp16x3_asm'][pi]
        assert torch.equal(a['z_samples'], b['z_samples']) and torch.equal(a['rgb0'], b['rgb0'])       # the coarse pass is the same launch
        d = {k: (a[k] - b[k]).abs().max().item() for k in ('rgb_map', 'acc_map', 'depth_map')}
        n5 = int(((a['rgb_map'] - b['rgb_map']).abs().max(-1)[0] > 3e-5).sum())
        print(f'pose {pi}: fp16_mix from fp16x3_asm over {H * H} rays: {d}; rays beyond 3e-5: {n5}')
        worst = max(worst, d['rgb_map'])
        assert d['rgb_map'] <= 5e-5 and d['acc_map'] <= 1e-4 and d['depth_map'] <= 6e-4
    print(f'per frame (extras copied): fp16x3_asm {times["fp16x3_asm"] * 1e3:.1f} ms, fp16_mix {times["fp16_mix"] * 1e3:.1f} ms; worst rgb {worst:.2e}')
    assert times['fp16_mix'] < 0.92 * times['fp16x3_asm']
    # the watch: a sample of a rendered frame against three passes on the same chain
    eng.set_precision(PRECISIONS['fp16_mix'])
    from efficient_nerf_amd.teacher import get_rays
    ro, rd = (t.reshape(-1, 3) for t in get_rays(H, H, WF.focal(), WF.pose(1)[:3, :4], device=eng.device))
    ok, d = eng.spot_check(ro, rd, eng.render_rays(ro, rd))
    print('spot check:', ok, d)
    assert ok and d['rgb_map'] <= eng.AUTO_MAX_DIFF_MIX
    assert eng.step_down() == 'fp16x3_asm' and eng.precision_coarse == PRECISIONS['fp16x3_asm'] == eng.precision
    eng.close()


def test_synthetic_teacher_keeps_its_single_pass(pkg):
    """the smooth synthetic teacher is not affected: `auto` still takes fp16x1, the first rung"""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    from efficient_nerf_amd import create_data as CD
    H = 200
    focal = O.focal_from_angle(H)
    eng = NeRFEngine(H, H, focal, precision=PRECISIONS['fp16x3']).load_state_dicts(O.make_teacher_state(1), O.make_teacher_state(2))
    assert CD.choose_precision_for_rand(eng, H, H, focal) == 'fp16x1'
    eng.close()


def test_rebalanced_fine_network_is_the_same_function_and_fits_the_bf6_range(pkg):
    """NeRFEngine.rebalance_fine: the activation maxima of the fine network measured with the library's fp32 layer kernels agree with the
    CPU oracle's, the power-of-two reparametrisation renders the same image in three passes (the same function in float32), every
    rescaled activation lies in (target / 2, target] (NeRFEngine.REBALANCE_TARGET), and the mixed chain is
    closer to three passes on the rebalanced network than on the original"""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    from efficient_nerf_amd.teacher import get_rays, rebalanced_state
    from oracle import whole_frame as WF
    import torch.nn.functional as F
    sds = WF.load_teacher()
    H = WF.H
    eng = NeRFEngine(H, H, WF.focal(), precision=PRECISIONS['fp16x3_asm']).load_state_dicts(*sds)
    ro, rd = (t.reshape(-1, 3) for t in get_rays(H, H, WF.focal(), WF.pose(0)[:3, :4], device=eng.device))
    idx = torch.arange(0, H * H, 40, device=ro.device)
    ros, rds = ro[idx].contiguous(), rd[idx].contiguous()
    ref = {k: v.clone() for k, v in eng.render_rays(ros, rds, extras=True).items()}
    mx = eng.fine_activation_maxima(ros, rds, ref['z_vals'], max_points=1 << 20)
    # the oracle's activations on the same points (CPU fp32)
    pts = (ros[:, None, :] + rds[:, None, :] * ref['z_vals'][:, :, None]).reshape(-1, 3).cpu()
    vd = (rds / rds.norm(dim=-1, keepdim=True))[:, None, :].expand(-1, ref['z_vals'].shape[1], 3).reshape(-1, 3).cpu()
    sd = sds[1]
    e, ev = O.nerf_embed(pts, 10), O.nerf_embed(vd, 4)
    h, want = e, {}
    with torch.no_grad():
        for i in range(8):
            h = F.relu(F.linear(h, sd[f'pts_linears.{i}.weight'], sd[f'pts_linears.{i}.bias']))
            want[f'h{i}'] = float(h.abs().max())
            if i == 4:
                h = torch.cat([e, h], -1)
        feat = F.linear(h, sd['feature_linear.weight'], sd['feature_linear.bias'])
        want['feature'] = float(feat.abs().max())
        want['views'] = float(F.relu(F.linear(torch.cat([feat, ev], -1), sd['views_linears.0.weight'], sd['views_linears.0.bias'])).abs().max())
    print('maxima (HIP fp32 layers):', {k: round(v, 2) for k, v in mx.items()})
    for k in want:
        assert abs(mx[k] - want[k]) <= 1e-3 * want[k], (k, mx[k], want[k])
    assert mx['views'] > 50 and mx['h7'] > 14            # beyond the chain's fixed range: what the rebalancing is for
    eng.set_precision(PRECISIONS['fp16_mix'])
    before = (eng.render_rays(ros, rds)['rgb_map'] - ref['rgb_map']).abs().max().item()
    sh = eng.rebalance_fine(ros, rds, ref['z_vals'])
    print('shifts:', sh)
    assert sh is not None and sh['h0'] == sh['h1'] == 0 and sh['views'] >= 2 and sh['h2'] < 0        # the tail down, the small front layers up
    for k, v in eng.fine_maxima.items():
        if k not in ('h0', 'h1'):
            assert eng.REBALANCE_TARGET / 2 < v / 2.0 ** sh[k] <= eng.REBALANCE_TARGET, (k, v, sh[k])
    after = (eng.render_rays(ros, rds)['rgb_map'] - ref['rgb_map']).abs().max().item()
    eng.set_precision(PRECISIONS['fp16x3_asm'])
    same = eng.render_rays(ros, rds, extras=True)
    d3 = (same['rgb_map'] - ref['rgb_map']).abs().max().item()
    print(f'three passes on the rebalanced network vs the original: rgb {d3:.1e}, raw {(same["raw"] - ref["raw"]).abs().max().item():.1e}; '
          f'fp16_mix from three passes on {idx.numel()} rays: {before:.2e} as loaded, {after:.2e} rebalanced')
    assert d3 <= 2e-6 and torch.equal(same['z_samples'], ref['z_samples'])
    assert after <= max(before, 3e-5) and after <= 5e-5
    eng.close()


@pytest.mark.parametrize('mode', ['fp16x3_asm', 'fp16_mix'])
def test_coarse_pass_without_its_view_branch_changes_nothing_but_rgb0(pkg, mode):
    """nerf_set_skip_rgb0 (round 6): with the coarse network in fp16x3_asm the render pipeline runs it without feature_linear /
    views_linears.0 / rgb_linear (model/nerf_raybased.py:391-398) -- the coarse densities, hence z_samples, hence every map of the fine
    pass, and acc0 / disp0 / z_std are BITWISE what they were; rgb0 is gone (and asked for through the C-ABI: refused); the frame is
    faster; nerf_run_network still evaluates the whole coarse network; a coarse mode without that build is unaffected"""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    from efficient_nerf_amd._lib import R2LError
    from oracle import whole_frame as WF
    sds = WF.load_teacher()
    H = WF.H
    eng = NeRFEngine(H, H, WF.focal(), precision=PRECISIONS[mode]).load_state_dicts(*sds)
    pose = WF.pose(1)

    def timed():
        eng.render(pose)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.render(pose)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3
    full = {k: v.clone() for k, v in eng.render(pose, extras=True).items()}
    t_full = timed()
    ro, rd = (t.reshape(-1, 3)[:4096].contiguous() for t in __import__('efficient_nerf_amd').teacher.get_rays(H, H, WF.focal(), pose[:3, :4], device=eng.device))
    raw0 = eng.run_network(0, ro, rd, eng.z_coarse.cuda()).clone()
    eng.set_skip_rgb0(True)
    got = eng.render(pose, extras=True)
    t_skip = timed()
    assert 'rgb0' not in got and set(got) == set(full) - {'rgb0'}
    for k in got:          # bit patterns: disp of an empty ray is 0 / 0 = NaN in the reference too (main.py:609-610)
        if k != 'raw':     # (the fine network's second exit leaves zero colours in raw where no density is positive: the next test)
            assert torch.equal(got[k].view(torch.int32), full[k].view(torch.int32)), k
    assert torch.equal(got['raw'][..., 3], full['raw'][..., 3])
    assert torch.equal(eng.run_network(0, ro, rd, eng.z_coarse.cuda()), raw0) and float(raw0[..., :3].abs().max()) > 0
    with pytest.raises(R2LError, match='rgb0 was not computed'):
        from efficient_nerf_amd._lib import lib, check, dptr, current_stream
        buf = torch.empty((16, 3), device='cuda')
        check(lib().nerf_copy_extras(eng._ctx, 16, dptr(buf), None, None, None, current_stream()))
    print(f'{mode}: {t_full * 1e3:.1f} ms per 400 x 400 frame, without the coarse view branch {t_skip * 1e3:.1f} ms ({(1 - t_skip / t_full) * 100:.1f} % less)')
    assert t_skip < 0.975 * t_full
    eng.set_precision(PRECISIONS['fp16x3'])              # the compiler-scheduled mode has no such build: nothing is skipped, rgb0 is there
    assert 'rgb0' in eng.render(pose, extras=True)
    eng.close()


@pytest.mark.parametrize('mode', ['fp16x3_asm', 'fp16_mix'])
def test_second_exit_behind_the_density_changes_no_map(pkg, mode):
    """nerf_set_skip_rgb0, fine network (round 6): the chain with the alpha row first and a second exit for workgroup tiles without a
    positive density (NERF_GEN_FMT=f16p3s / mixs).  On the trained-like teacher (three quarters of the rays see nothing) every map and
    extra is BITWISE what the full chain gives -- a sample with density <= 0 has weight 0 exactly (main.py:600-606) --, `raw` keeps every
    density and shows zero colours exactly on tiles without a positive one, and the frame is faster.  With density noise the full chain
    runs (the noise can lift a density above zero)."""
    from efficient_nerf_amd import NeRFEngine, PRECISIONS
    from oracle import whole_frame as WF
    sds = WF.load_teacher()
    H = WF.H
    eng = NeRFEngine(H, H, WF.focal(), precision=PRECISIONS[mode]).load_state_dicts(*sds)

    def timed(pose):
        eng.render(pose)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.render(pose)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 3
    for pi in (0, 1):
        pose = WF.pose(pi)
        eng.set_skip_rgb0(False)
        full = {k: v.clone() for k, v in eng.render(pose, extras=True).items()}
        t_full = timed(pose)
        eng.set_skip_rgb0(True)
        got = eng.render(pose, extras=True)
        t_skip = timed(pose)
        for k in got:
            if k != 'raw':
                assert torch.equal(got[k].view(torch.int32), full[k].view(torch.int32)), k
        raw, raw_f = got['raw'].reshape(-1, 4), full['raw'].reshape(-1, 4)
        assert torch.equal(raw[:, 3], raw_f[:, 3])                                   # every density as before
        n = raw.shape[0] // 128 * 128
        dead = ~(raw_f[:n, 3] > 0).reshape(-1, 128).any(-1)                          # workgroup tiles of the fine launch without a positive density
        tiles, tiles_f = raw[:n, :3].reshape(-1, 128, 3), raw_f[:n, :3].reshape(-1, 128, 3)
        assert not tiles[dead].any() and torch.equal(tiles[~dead], tiles_f[~dead])   # zero colours exactly there, the full chain's everywhere else
        print(f'{mode} pose {pi}: {float(dead.float().mean()):.3f} of the fine launch\'s {dead.numel()} tiles take the second exit; '
              f'{t_full * 1e3:.1f} -> {t_skip * 1e3:.1f} ms per frame ({(1 - t_skip / t_full) * 100:.1f} % less, incl. the coarse view branch)')
        assert float(dead.float().mean()) > 0.4 and t_skip < 0.95 * t_full
    # density noise: the full chain
    ro, rd = (t.reshape(-1, 3)[:8192].contiguous() for t in __import__('efficient_nerf_amd').teacher.get_rays(H, H, WF.focal(), WF.pose(0)[:3, :4], device=eng.device))
    a = eng.render_rays(ro, rd, extras=True, raw_noise_std=1.0, pytest=True)['raw']
    eng.set_skip_rgb0(False)
    b = eng.render_rays(ro, rd, extras=True, raw_noise_std=1.0, pytest=True)['raw']
    assert torch.equal(a, b)
    eng.close()
