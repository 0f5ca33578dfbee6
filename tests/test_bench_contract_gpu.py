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
                        '--cpu-rays', '8000', '--no-teacher'], cwd=ROOT, capture_output=True, text=True, timeout=900)
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
    # the middle rung of the ladder on the networks it is for
    em = d['e4m3_mode']
    assert em['auto_precision'] == 'fp16_e4m3' and em['max_act_exponent'] == 4 and em['linf_vs_cpu_oracle'] <= 1e-4
    # the stress weights of SURVEY 8(d) must not be rendered with the bf6 terms: the library's own range check decides
    sw = d['stress_weights']
    assert sw['auto_precision'] == 'fp16x3_asm' and sw['max_act_exponent'] > 3 and sw['linf_vs_cpu_oracle'] <= 1e-4
    assert d['calibration']['max'] <= d['calibration']['auto_precision_limit']     # ... and would keep fp16_fp8 for the standard set
