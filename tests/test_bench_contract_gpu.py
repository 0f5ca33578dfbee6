"""GPU: bench.py prints exactly one JSON line carrying the driver's contract fields, the roofline
and (at N = 1) the CPU baseline + parity objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_schema():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--cpu-rays', '8000', '--no-teacher', '--no-trained-like'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'rays/s' and d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1
    assert d['higher_is_better'] is True and d['scaling'] == 'weak' and d['vs_baseline'] is None
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 640000 * 1e3 / d['ms_per_step']) <= 1e-6 * d['value']
    rf = d['roofline']
    assert rf['bound'] == 'mfma' and rf['unit'] == 'TFLOP/s' and rf['peak'] == 2500.0
    assert abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-12 and 0.05 < rf['frac'] < 1.0
    # the dominant kernel of the default mode is r2l_body_kernel: the 86 body layers and the fused tail layer,
    # 2 * (86 * 256^2 + 3 * 256) flop per ray
    assert rf['kernel'] == 'r2l_body_kernel' and rf['algorithmic_flops_per_ray'] == 2 * (86 * 65536 + 768)
    assert abs(rf['achieved'] - 2 * (86 * 65536 + 768) * 640000 / (rf['avg_kernel_ms'] * 1e-3) / 1e12) <= 1e-6 * rf['achieved']
    wp = rf['whole_path']
    assert wp['algorithmic_flops_per_ray'] == 11789824 and wp['frac'] <= rf['frac']
    assert abs(wp['achieved'] - 11789824 * 640000 / (d['ms_per_step'] * 1e-3) / 1e12) <= 1e-6 * wp['achieved']
    cb = d['cpu_baseline']
    assert cb['kind'] == 'port' and cb['unit'] == 'rays/s' and cb['cores'] >= 1 and cb['value'] > 0 and 'sample' in cb
    assert d['parity']['within_tolerance'] and d['parity']['linf_vs_cpu_oracle'] <= 1e-4
    # SURVEY 8(d)'s protocol: per-step device times beside the mean, L_inf over >= 3 frames (one of them the pose the
    # exponents were measured on, the others not), north_star's second tolerance (PSNR delta < 0.01 dB)
    assert 0 < d['min_ms'] <= d['median_ms'] <= d['max_ms'] and d['median_ms'] <= 1.05 * d['ms_per_step']
    fr = d['parity']['frames']
    assert len(fr) >= 3 and len({f['pose'] for f in fr}) == len(fr) and any(not f['calibration_pose'] for f in fr)
    assert all(f['linf'] <= 1e-4 and f['rays'] >= 2400 for f in fr) and d['parity']['rays_checked'] == sum(f['rays'] for f in fr)
    assert d['parity']['psnr_delta_db'] < 0.01
    # every ray of the timed steps was watched (h0; the guarded launches among them: every operand set): nothing clamped
    rw = d['range_watch']
    assert rw['saturated'] == 0 and rw['launches'] == 2 and 0 < rw['h0_fill'] < 1 and rw['worst_fill'] < 1
    assert d['calibration']['measured_on'].startswith('every ray')
    # `auto` on the headline's own weights: the rung the limits name, verified against three passes
    aw = d['auto_on_these_weights']
    assert aw['precision'] == 'fp16_fp8' and 0 < aw['rgb_diff_from_three_passes_on_the_probe_frame'] <= aw['limit'] <= 1e-4
    # the middle rung of the ladder on the networks it is for
    em = d['e4m3_mode']
    assert em['auto_precision'] == 'fp16_e4m3' and em['max_act_exponent'] == 4 and 8 < em['max_abs_activation'] <= 10 and em['linf_vs_cpu_oracle'] <= 1e-4
    # the stress weights of SURVEY 8(d) must not be rendered with the bf6 terms throughout: the library's own range check decides
    # (since round 5 the split rung: as many leading blocks in three passes as its measurement asks for, or all of them)
    sw = d['stress_weights']
    assert sw['auto_precision'] in ('fp16_split', 'fp16_split8', 'fp16x3_asm') and sw['max_act_exponent'] > 3 and sw['linf_vs_cpu_oracle'] <= 1e-4
    assert d['calibration']['max'] <= d['calibration']['auto_precision_limit']     # ... and would keep fp16_fp8 for the standard set


def test_bench_line_round4_fields():
    """the config-5 leg at the reference's own group size, where the traffic figure comes from,
    and for which activations the headline rate holds"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--no-cpu-baseline', '--no-teacher'], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert 'traffic_source' in d['roofline'] and isinstance(d['roofline']['traffic_source'], str)
    assert (d['roofline']['traffic'] is None) == ('not measured in this run' not in d['roofline']['traffic_source'])
    assert d['value_valid_for'].startswith('activation exponent <= 3')
    cd = d['create_data']
    assert cd['poses'] == 200 and cd['groups'] == 2 and cd['shards'] == 2 * (100 * 160000 // 4096) == 7812
    assert cd['shard_bytes_total'] == 7812 * (128 + 4096 * 36)
    # 2 MLP launches per pose + 2 per spot check of the watch (one check per save group while the mode is a fast one)
    assert cd['watch']['checks'] == 2 and cd['watch']['fallbacks'] == [] and cd['watch']['precision'] == cd['precision'] == 'fp16x1'
    assert cd['watch']['worst']['rgb_map'] <= 3e-5
    assert abs(cd['poses_per_s'] - 200 / cd['wall_s']) < 1e-9 and cd['mlp_launches'] == 400 + 2 * cd['watch']['checks']
    # the teacher's MLP launches are the leg: >= 90 % of its wall clock (VERDICT r3's bar; one group alone measured 0.94-0.96 on
    # boxes whose file system creates the 3,906 files in 0.2-0.3 s and 0.85-0.94 on one where that took up to 0.9 s: the exposed
    # tail of the LAST group -- shuffle gather, copy, file writes -- is paid once per job, here once per two groups)
    assert cd['mlp_kernel_share_of_wall'] >= 0.90, cd
    assert cd['tail_s'] < 1.5 and cd['extrapolated_n_pose_kd_10000_hours_one_gpu'] < 1.0 < cd['reference_quotes_hours']
    # the trained-like fixture's rungs and rates in the line (CPU-oracle fields only with the CPU baseline on)
    tl = d['trained_like']
    st = tl['student']
    assert st['rung'] in ('fp16_split', 'fp16_split8') and st['max_abs_activation'] > 10 and st['rays_per_s'] > st['rays_per_s_fp16x3_asm'] > 1e7
    assert 0 <= st['split_block'] < st['n_block'] == 43 and st['watch_worst_rgb_diff_from_three_passes'] <= st['watch_limit']
    # round 6: the trained-like teacher gets fp16_mix (coarse three passes, fine with its first two layers in three passes), measured and watched
    assert tl['teacher']['precision'] == 'fp16_mix' and tl['teacher']['probe_diffs_from_fp16x3']['fp16x1'] > 1e-3
    assert tl['teacher']['probe_diffs_from_fp16x3']['fp16_mix'] <= tl['teacher']['limits']['fp16_mix'] == 5e-5
    assert tl['teacher']['probe_diffs_from_fp16x3']['fp16x3_asm'] <= tl['teacher']['limits']['fp16x3_asm']
    assert tl['teacher']['frac_of_fp16_mfma_peak'] > 0.22
    assert tl['create_data']['precision'] == 'fp16_mix' and tl['create_data']['poses'] == 100 and tl['create_data']['shards'] == 3906 and tl['create_data']['poses_per_s'] > 8
    assert tl['create_data']['watch']['checks'] >= 1 and tl['create_data']['watch']['fallbacks'] == []
    ro = st['roofline']          # the student's kernels timed with HIP events, as the headline's
    assert ro['launches'] == 10 + ro['rerenders'] and 0 < ro['avg_kernel_ms'] < st['ms_per_frame'] and 0.2 < ro['frac'] < 0.4
    assert tl['teacher']['mlp_launches'] == 6 and 0 < tl['teacher']['mlp_kernel_ms_per_frame'] <= tl['teacher']['ms_per_frame']
    assert tl['teacher']['whole_frame_rgb_linf_from_fp16x3']['fp16x1'] > 1e-3
    assert 'value_valid_for' in d and 'trained-like' in d['value_valid_for']


def _torchrun_bench(extra, port):
    env = dict(os.environ, R2L_DIST_BACKEND='gloo')
    return subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                           '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2',
                           '--warmup', '1', '--no-cpu-baseline', '--no-teacher', '--no-trained-like'] + extra, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=900)


def test_scaling_run_refuses_a_silent_collective_fallback():
    """at --gpus N > 1 the record must time r2l_gather_image or say loudly that it does not.  Two ranks with
    gloo between them (both on this GPU: the library's RCCL collective cannot run there): without --allow-fallback bench.py
    exits non-zero and prints no JSON line; with it the line names the stand-in and carries gather_check."""
    import socket

    def port():
        s = socket.socket()
        s.bind(('127.0.0.1', 0))
        p = s.getsockname()[1]
        s.close()
        return p

    r = _torchrun_bench([], port())
    assert r.returncode != 0 and 'r2l_gather_image (RCCL, C-ABI) cannot assemble' in r.stderr, r.stdout[-800:] + r.stderr[-1500:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    r = _torchrun_bench(['--allow-fallback'], port())
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-1500:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert d['n_gpus'] == 2 and 'torch.distributed all_gather (gloo)' in d['config']['gather']
    gc = d['gather_check']
    assert gc['assembled_frame_equals_own_render_on_every_rank'] is True and gc['fallback_allowed'] is True
    assert gc['collective'] == d['config']['gather']


def _self_launched_bench(extra):
    env = dict(os.environ, R2L_DIST_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                           '--no-cpu-baseline', '--no-teacher', '--no-create-data', '--no-trained-like', '--launch-timeout', '800'] + extra, cwd=ROOT,
                          env=env, capture_output=True, text=True, timeout=900)


def test_bench_starts_its_own_ranks_when_called_as_the_driver_calls_it():
    """the driver's command is `python3 bench.py --gpus N`, no torchrun.  bench.py then starts the N ranks
    itself (efficient-nerf_amd/launch.py: fresh child processes, before any GPU call) and relays rank 0's line.  Two ranks with
    gloo between them on this one GPU: with --allow-fallback one JSON line with n_gpus 2 and gather_check; without it exit
    code 3 (the scaling record times r2l_gather_image or nothing) and no line."""
    r = _self_launched_bench(['--allow-fallback'])
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['config']['frames_per_step'] == 2
    assert abs(d['value'] - 2 * 640000 * 1e3 / d['ms_per_step']) <= 1e-6 * d['value']
    gc = d['gather_check']
    assert gc['assembled_frame_equals_own_render_on_every_rank'] is True and gc['fallback_allowed'] is True
    r = _self_launched_bench([])
    assert r.returncode == 3, (r.returncode, r.stdout[-800:] + r.stderr[-1500:])
    assert 'r2l_gather_image (RCCL, C-ABI) cannot assemble' in r.stderr and '[launch] rank' in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
