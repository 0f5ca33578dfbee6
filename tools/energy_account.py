#!/usr/bin/env python
"""Energy account of the two generated kernels: prices the instruction mix of r2l_body_kernel (per ResMLP
block and wave) and nerf_chain_kernel (per 128-point tile and wave) with the joules per instruction class that
tools/energy_probe measured on the SAME box under the same power cap, and compares the sum with the package power read while the
real kernels ran (tools/energy_run.sh).

    python tools/energy_account.py gpurun_out/r04_energy > profiles/r04_energy_account.txt

Model: P = P_static + P_clk * f / f_max + sum over classes (events / s) x (J / event).
  P_static           idle package power (no kernel)
  P_clk              spin (all 1,024 waves looping, nothing else) minus idle, at f_max
  J / event          single-class rows: (P - P_spin) / (events / s); those rows run uncapped at f_max.  MFMA rows run at the
                     cap: (P - P_static - P_clk f / f_max) / (events / s) at the clock the row ran at.
The instruction counts come from the committed .inc files (the block loop of r2l_body_asm.inc, the tile block of
nerf_mlp_asm.inc); events / s = count / (time per block or tile), from the kernels' HIP-event times."""
import collections
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc')
WAVES = 1024
KEYS = 'M H B K R W G L C X P F A'.split()


def parse_probe(path):
    rows, idle = [], None
    for ln in open(path):
        if ln.startswith('idle'):
            idle = float(ln.split('power')[1].split('W')[0])
            continue
        m = re.match(r'(.*?)(\s+M\s+\d+ H.*)iter/s/wave ([0-9.e+]+)\s+power\s+([0-9.]+) W\s+sclk\s+([0-9.]+) MHz', ln)
        if not m:
            continue
        tag, mid, it, p, f = m.group(1).strip(), m.group(2), float(m.group(3)), float(m.group(4)), float(m.group(5)) / 1e3
        n = {k: int(v) for k, v in re.findall(r'\b([MHBKRWGLCXPFA])\s+(\d+)', mid)}
        rows.append(dict(tag=tag, n=n, it=it, p=p, f=f, lockstep='lockstep' in mid))
    return idle, rows


def counts(path, start=None, end=None):
    lines = [ln.strip().strip('"').replace('\\n\\t', '') for ln in open(path) if ln.startswith('"')]
    if start:
        i = [k for k, ln in enumerate(lines) if ln.startswith(start)][0]
        j = [k for k, ln in enumerate(lines) if end in ln and k > i][-1]
        lines = lines[i:j]
    return collections.Counter(ln.split()[0] for ln in lines if ln and not ln.endswith(':'))


def classes(c, shapes):
    """instruction counter -> events per class of the probe"""
    f16 = c['v_mfma_f32_32x32x16_f16'] + c['v_mfma_f32_16x16x32_f16']
    b6 = c['v_mfma_scale_f32_32x32x64_f8f6f4'] + c['v_mfma_scale_f32_16x16x128_f8f6f4']
    out = {('M' if shapes == 32 else 'H'): f16, ('B' if shapes == 32 else 'K'): b6,
           'R': c['ds_read_b128'] + 0.5 * c['ds_read_b64'] + 0.75 * c['ds_read_b96'],     # in KiB-sized reads
           'G': c['global_load_lds_dwordx4'], 'C': c['v_cvt_scalef32_pk32_bf6_f16'],
           'X': 0.5 * (c['v_fma_mixlo_f16'] + c['v_fma_mixhi_f16']), 'P': c['v_cvt_pk_f16_f32'],
           'A': 0.5 * (c['v_accvgpr_write_b32'] + c['v_accvgpr_read_b32'])}
    other_valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith(('v_mfma', 'v_cvt_scalef32_pk32', 'v_fma_mix',
                                                                                        'v_cvt_pk_f16', 'v_accvgpr')))
    out['F'] = other_valu        # v_max, address arithmetic: priced as plain fp32 VALU
    return out


def main():
    d = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r04_energy')
    idle, rows = parse_probe(os.path.join(d, 'probe.txt'))
    by = {r['tag']: r for r in rows}
    spin = by['spin (loop overhead only)']
    fmax = spin['f']
    p_clk = spin['p'] - idle
    base = lambda f: idle + p_clk * f / fmax      # noqa: E731
    print('tools/energy_account.py on %s (one MI355X, one gpurun call: probe, then the two kernels under rocm-smi)' % os.path.relpath(d, ROOT))
    print()
    print('1. Package power without arithmetic: idle %.0f W; 1,024 waves spinning at %.2f GHz %.0f W  ->  P_static %.0f W + %.0f W x f / %.2f GHz'
          % (idle, fmax, spin['p'], idle, p_clk, fmax))
    print()
    print('2. Joules per wave-level instruction (64 lanes), from single-class loops (uncapped, %.2f GHz) and MFMA loops (at the cap):' % fmax)
    single = [('F', 'v_fma_f32 x64', 'v_fma_f32 (plain fp32 VALU)'), ('P', 'v_cvt_pk_f16_f32 x32', 'v_cvt_pk_f16_f32'),
              ('X', 'v_fma_mix pair x32', 'v_fma_mixlo + v_fma_mixhi pair'), ('C', 'cvt_pk32_bf6 x8', 'v_cvt_scalef32_pk32_bf6_f16 (32 values / lane)'),
              ('A', 'accvgpr pair x32', 'v_accvgpr_write + read pair'), ('R', 'ds_read_b128 x16', 'ds_read_b128 (1 KiB)'),
              ('W', 'ds_write_b128 x16 (random data)', 'ds_write_b128 (1 KiB)'), ('Gm', 'LDS-DMA x16 (Infinity Cache)', 'LDS-DMA 1 KiB, every wave its own lines (Infinity Cache)'),
              ('G', 'LDS-DMA x4 (L2, lockstep)', 'LDS-DMA 1 KiB, all CUs the same lines in step (L2), 0.25 of the path'),
              ('Gs', 'LDS-DMA x16 (L2, lockstep)', 'LDS-DMA 1 KiB, the same, path saturated'),
              ('L', 'global_load x16 (L2, lockstep)', 'global_load_dwordx4 1 KiB (L2), path saturated'),
              ('M', 'mfma 32x32x16 f16 x16', 'v_mfma_f32_32x32x16_f16 (16,384 MAC; weights N(0,1), activations relu(N(0,1)))'),
              ('H', 'mfma 16x16x32 f16 x32', 'v_mfma_f32_16x16x32_f16 (8,192 MAC)'),
              ('B', 'mfma 32x32x64 bf6 x16', 'v_mfma_scale_f32_32x32x64_f8f6f4 bf6 x bf6 (65,536 MAC)'),
              ('K', 'mfma 16x16x128 bf6 x32', 'v_mfma_scale_f32_16x16x128_f8f6f4 bf6 x bf6 (32,768 MAC)')]
    E = {}
    for key, tag, what in single:
        r = by[tag]
        k0 = key[0]
        rate = r['it'] * r['n'][k0] * WAVES
        e = (r['p'] - base(r['f'])) / rate
        E[key] = e
        per = ''
        if k0 in 'MHBK':
            macs = {'M': 16384, 'H': 8192, 'B': 65536, 'K': 32768}[k0]
            per = '  = %.2f pJ / MAC, %.3e MAC/s at %.2f GHz sclk, %.0f W' % (e / macs * 1e12, rate * macs, r['f'], r['p'])
        elif k0 in 'RWGL':
            per = '  = %.1f pJ / B at %.1f TB/s, %.0f W' % (e / 1024 * 1e12, rate * 1024 / 1e12, r['p'])
        print('   %-2s %-86s %7.2f nJ%s' % (key, what, e * 1e9, per))
    print('   16x16 against 32x32 shapes, energy per MAC: fp16 %.0f %%, bf6 %.0f %%' % (100 * (2 * E['H'] / E['M'] - 1), 100 * (2 * E['K'] / E['B'] - 1)))
    print()
    print('3. Does the sum of the parts predict a mix?  (rows of the same probe at the cap; predicted with the energies above)')
    for tag in ('mfma 32x32 f16:bf6 2:1', 'mfma 16x16 f16:bf6 2:1', 'mfma16 + ds_read 1.4/MFMA', 'mfma16 + VALU 2/MFMA', 'body replica, no DMA'):
        r = by[tag]
        dyn = sum(E[k] * n for k, n in r['n'].items() if n and k in E) * r['it'] * WAVES
        print('   %-34s measured %6.0f W at %.2f GHz   predicted %6.0f W (%+.1f %%)' % (tag, r['p'], r['f'], base(r['f']) + dyn,
                                                                                        100 * ((base(r['f']) + dyn) / r['p'] - 1)))
    # ---- the kernels -----------------------------------------------------------------------------------------------
    pw = collections.defaultdict(list)
    for ln in open(os.path.join(d, 'power.txt')):
        m, mp, mf = re.match(r'(body|chain) ', ln), re.search(r'Power \(W\): ([0-9.]+)', ln), re.search(r'\(([0-9]+)Mhz\)', ln)
        if m and mp and mf:
            pw[m.group(1)].append((float(mp.group(1)), float(mf.group(1)) / 1e3))
    body_ms = float(re.search(r'timed kernel mean ([0-9.]+) ms', open(os.path.join(d, 'body.txt')).read()).group(1))
    chain_ms = float(re.search(r'teacher fp16_fp8 \S+: ([0-9.]+) ms/frame', open(os.path.join(d, 'teacher.txt')).read()).group(1))
    kern = [('r2l_body_kernel', 'body', classes(counts(os.path.join(CSRC, 'r2l_body_asm.inc'), 'L_block_', 's_cbranch_scc1 L_block_'), 32),
             body_ms * 1e-3 / (20 * 43), 'ResMLP block (2 layers) of a 128-ray tile, per wave: %.3f ms per launch / (20 tiles per workgroup x 43 blocks)' % body_ms),
            ('nerf_chain_kernel', 'chain', classes(counts(os.path.join(CSRC, 'nerf_mlp_asm.inc')), 16),
             chain_ms * 1e-3 * 0.994 / (400 * 400 * 256 / 128 / 256), '128-point tile (11 layers), per wave: %.1f ms per 400x400 frame x 0.994 / 1,250 tiles per CU' % chain_ms)]
    names = {'M': 'fp16 MFMA 32x32x16', 'H': 'fp16 MFMA 16x16x32', 'B': 'bf6 MFMA 32x32x64', 'K': 'bf6 MFMA 16x16x128', 'R': 'ds_read (KiB)',
             'G': 'LDS-DMA (KiB, L2)', 'C': 'v_cvt_scalef32_pk32', 'X': 'v_fma_mix pairs', 'P': 'v_cvt_pk_f16_f32', 'A': 'v_accvgpr pairs',
             'F': 'other VALU'}
    for n, (kname, key, cl, T, what) in enumerate(kern):
        p_meas = sum(p for p, _ in pw[key]) / len(pw[key])
        f_meas = sum(f for _, f in pw[key]) / len(pw[key])
        print()
        print('%d. %s: %s = %.2f us; rocm-smi while it runs: %.0f W at sclk %.2f GHz' % (4 + n, kname, what, T * 1e6, p_meas, f_meas))
        tot = 0
        parts = []
        for k in ('M', 'H', 'B', 'K', 'R', 'G', 'C', 'X', 'P', 'A', 'F'):
            if not cl.get(k):
                continue
            e = cl[k] * E[k]
            tot += e
            parts.append((k, cl[k], e))
        stat = base(f_meas)
        dyn_w = tot * WAVES / T
        for k, cnt, e in parts:
            print('     %-22s %7.1f x %6.2f nJ = %8.0f nJ   %5.1f %% of the dynamic energy   %6.0f W' % (names[k], cnt, E[k] * 1e9, e * 1e9, 100 * e / tot,
                                                                                                      e * WAVES / T))
        print('     %-22s %37.0f nJ   per wave and unit             %6.0f W' % ('dynamic, sum', tot * 1e9, dyn_w))
        print('     %-22s %79.0f W   (%.0f W + %.0f W x %.2f / %.2f GHz)' % ('static + clock', stat, idle, p_clk, f_meas, fmax))
        print('     %-22s %79.0f W   measured %.0f W: %+.1f %%' % ('sum', stat + dyn_w, p_meas, 100 * ((stat + dyn_w) / p_meas - 1)))
    # ---- the levers ------------------------------------------------------------------------------------------------
    idle2, rows2 = parse_probe(os.path.join(d, 'probe_ab.txt'))
    ab = {r['tag']: r for r in rows2}
    print()
    print('6. Levers, ranked by the dynamic energy they touch (body kernel), and what a replica of the body mix measures for them.')
    print('   Replicas: the same loop with the body kernel\'s instruction ratios per 16 MFMA slots, loads in flight across the loop edge, all at')
    print('   the cap; iter/s is throughput of equal work (same MACs, same LDS / L2 bytes, same VALU):')
    for tag in ('body replica (32x32 shapes)', 'body replica (32x32 shapes) again', 'body replica in 16x16 shapes', 'body replica in 16x16 shapes again',
                'body replica, f16 16x16 + bf6 32x32', 'body replica (32x32), no DMA', 'body replica (16x16), no DMA', 'B-from-LDS variant (32x32)',
                'B-from-LDS variant (16x16)', 'chain replica (16x16)', 'chain, weights in registers'):
        r = ab[tag]
        print('     %-36s %.4e iter/s/wave   %6.0f W   sclk %.2f GHz' % (tag, r['it'], r['p'], r['f']))
    a32 = 0.5 * (ab['body replica (32x32 shapes)']['it'] + ab['body replica (32x32 shapes) again']['it'])
    a16 = 0.5 * (ab['body replica in 16x16 shapes']['it'] + ab['body replica in 16x16 shapes again']['it'])
    print('   16x16 shapes for all MFMAs: %+.1f %% (the per-MAC energies of section 2 promise %+.1f %% of the body\'s dynamic energy);'
          % (100 * (a16 / a32 - 1), 100 * (kern[0][2]['M'] * (2 * E['H'] - E['M']) + kern[0][2]['B'] * (2 * E['K'] - E['B'])) /
             sum(kern[0][2][k] * E[k] for k in kern[0][2] if k in E)))
    print('   activations through LDS, weights from L2 into registers (VERDICT r3 4c): %+.1f %% (32x32), chain %+.1f %%'
          % (100 * (ab['B-from-LDS variant (32x32)']['it'] / a32 - 1), 100 * (ab['chain, weights in registers']['it'] / ab['chain replica (16x16)']['it'] - 1)))
    r16, r32 = ab['body replica (16x16), no DMA'], ab['body replica (32x32), no DMA']
    e_other = sum(E[k] * n for k, n in r16['n'].items() if n and k in E and k not in 'MHBK')
    imp = lambda r: (r['p'] - (idle + p_clk * r['f'] / fmax)) / (r['it'] * WAVES) - e_other      # noqa: E731
    print('   why the shapes deliver less than their isolated energies: MFMA energy per 16 slots implied by the no-DMA replicas (measured power')
    print('   minus static, clock and the other classes): 32x32 %.0f nJ (isolated: %.0f), 16x16 %.0f nJ (isolated: %.0f) -- between other instructions'
          % (imp(r32) * 1e9, (11 * E['M'] + 5 * E['B']) * 1e9, imp(r16) * 1e9, (22 * E['H'] + 10 * E['K']) * 1e9))
    print('   a 16-cycle MFMA fetches its operands twice as often per MAC as a 32-cycle one, and that delivery is what the isolated loop (two operand')
    print('   pairs, nothing in between) gets for free.')
    print()
    mix = by['mfma 32x32 f16:bf6 2:1']
    t_min = 384 / 16 / mix['it']
    T = kern[0][3]
    print('7. What this package gives the arithmetic at most: the block\'s 256 + 128 MFMAs alone, back to back at the cap (row "mfma 32x32 f16:bf6')
    print('   2:1": %.0f W, %.2f GHz): %.2f us per block against the kernel\'s %.2f us -> the kernel runs at %.2f of the package\'s own MFMA-only rate'
          % (mix['p'], mix['f'], t_min * 1e6, T * 1e6, t_min / T))
    print('   for this mix; data movement and conversions (LDS reads %.0f %%, L2 -> LDS %.0f %%, VALU %.0f %% of the dynamic energy) are the rest.'
          % tuple(100 * x / sum(kern[0][2][k] * E[k] for k in kern[0][2] if k in E) for x in
                  (kern[0][2]['R'] * E['R'], kern[0][2]['G'] * E['G'], sum(kern[0][2][k] * E[k] for k in 'CXPAF'))))


if __name__ == '__main__':
    main()
