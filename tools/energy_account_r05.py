#!/usr/bin/env python
"""Round-5 energy account of the kernels that TRAINED-LIKE weights end on -- r2l_bodyx_kernel (the student's rung
fp16x3_asm), the teacher's three-pass chain (fp16x3_asm, nerf_chain_kernel<false, 2, true>) -- and of the single-pass chain with its
embedding in the stream (nerf_chain_emb_kernel).  Instruction counts from the committed .inc files; joules per wave-level
instruction as measured in round 4 on the same package class (profiles/r04_energy_account.txt section 2: tools/energy_probe under the
1,400 W cap); times from this round's runs (arguments).  Prints, per kernel, where the dynamic energy goes, the package power the
parts add up to at the measured time, and the time the MFMAs alone would take at the cap ("the package's own rate for this arithmetic").

    python tools/energy_account_r05.py > profiles/r05_energy_account.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc')
E = {'M': 15.00, 'H': 6.33, 'R': 2.66, 'G': 11.03, 'X': 1.41, 'P': 0.23, 'A': 0.63, 'F': 0.38, 'T': 1.5,   # nJ (r04 section 2); T: v_sin / v_rcp / v_sqrt, priced as four plain VALU
     # round 5 (profiles/r05_energy_probe_e4m3.txt): e4m3 x e4m3 MFMA 32x32x64 alone 1.868e6 x 16 per wave and second at 1,314 W and
     # 2,075 MHz -> (1,314 - 295 - 156 x 2.075 / 2.4) W / 3.06e10 per second = 28.9 nJ (0.44 pJ / MAC; bf6 x bf6: 19.84 nJ, 0.30);
     # v_cvt_scalef32_pk_fp8_f16 (TWO values per instruction) x 64: 592 W at 2,394 MHz -> 0.53 nJ
     'E': 28.9, 'Q': 0.53}
IDLE, PCLK, FMAX, CAP, WAVES = 295.0, 156.0, 2.40, 1300.0, 1024        # W, W at 2.40 GHz, GHz; ~1,300 W is what rocm-smi reads under the 1,400 W cap


def counts(path, start=None, end=None):
    lines = [ln.strip().strip('"').replace('\\n\\t', '') for ln in open(path) if ln.startswith('"')]
    if start:
        i = [k for k, ln in enumerate(lines) if ln.startswith(start)][0]
        j = [k for k, ln in enumerate(lines) if end in ln and k > i][-1]
        lines = lines[i:j]
    return collections.Counter(ln.split()[0] for ln in lines if ln and not ln.endswith(':'))


def classes(c):
    out = {'M': c['v_mfma_f32_32x32x16_f16'], 'H': c['v_mfma_f32_16x16x32_f16'],
           'R': c['ds_read_b128'] + 0.5 * c['ds_read_b64'], 'G': c['global_load_lds_dwordx4'],
           'X': 0.5 * (c['v_fma_mixlo_f16'] + c['v_fma_mixhi_f16']), 'P': c['v_cvt_pk_f16_f32'],
           'A': 0.5 * (c['v_accvgpr_write_b32'] + c['v_accvgpr_read_b32']),
           'T': c['v_sin_f32'] + c['v_rcp_f32_e32'] + c['v_sqrt_f32_e32'],
           'E': c['v_mfma_scale_f32_32x32x64_f8f6f4'], 'Q': c['v_cvt_scalef32_pk_fp8_f16']}
    out['F'] = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith(('v_mfma', 'v_fma_mix', 'v_cvt_pk_f16', 'v_cvt_scalef32_pk_fp8', 'v_accvgpr', 'v_sin', 'v_rcp', 'v_sqrt')))
    return out


NAMES = {'M': 'fp16 MFMA 32x32x16', 'H': 'fp16 MFMA 16x16x32', 'R': 'ds_read (KiB)', 'G': 'LDS-DMA (KiB, L2)', 'X': 'v_fma_mix pairs',
         'P': 'v_cvt_pk_f16_f32', 'A': 'v_accvgpr pairs', 'F': 'other VALU', 'T': 'transcendentals', 'E': 'e4m3 MFMA 32x32x64', 'Q': 'v_cvt_scalef32_pk_fp8'}


def account(title, cl, t_unit, what):
    print(title)
    print('   unit: %s = %.2f us' % (what, t_unit * 1e6))
    tot = sum(cl[k] * E[k] for k in cl if cl[k])
    for k in 'MHERGXPAFTQ':
        if cl.get(k):
            print('     %-22s %8.1f x %6.2f nJ = %8.0f nJ   %5.1f %% of the dynamic energy   %5.0f W' %
                  (NAMES[k], cl[k], E[k], cl[k] * E[k], 100 * cl[k] * E[k] / tot, cl[k] * E[k] * 1e-9 * WAVES / t_unit))
    dyn_w = tot * 1e-9 * WAVES / t_unit
    # the clock that makes static + clock + dynamic meet the cap
    print('     dynamic, sum            %35.0f nJ   per wave and unit            %5.0f W  (+ %3.0f W static + clock tree at ~1.8 GHz = %4.0f W; the cap reads ~%d W)'
          % (tot, dyn_w, IDLE + PCLK * 1.8 / FMAX, dyn_w + IDLE + PCLK * 1.8 / FMAX, CAP))
    mf = sum(cl[k] * E[k] for k in 'MHE' if cl.get(k))
    t_floor = mf * 1e-9 * WAVES / (CAP - IDLE - PCLK * 2.0 / FMAX)
    print('     the MFMAs alone at the cap: %.2f us per unit -> the kernel runs at %.2f of the package\'s own rate for this arithmetic;'
          % (t_floor * 1e6, t_floor / t_unit))
    print('     data movement (LDS reads %.0f %%, L2 -> LDS %.0f %%) and VALU (%.0f %%) are the rest of the joules' %
          (100 * cl['R'] * E['R'] / tot, 100 * cl['G'] * E['G'] / tot, 100 * sum(cl[k] * E[k] for k in 'XPAFTQ' if cl.get(k)) / tot))
    print()


def main():
    a = dict(x.split('=') for x in sys.argv[1:])
    bodyx_ms = float(a.get('bodyx_ms', 15.29))       # r2l_bodyx_kernel per 800 x 800 frame (profiles/r04_bench_n1_fp16x3_asm.json)
    p3_ms = float(a.get('p3_ms', 98.8))              # teacher fp16x3_asm MLP kernels per 400 x 400 frame (profiles/r05_teacher_x3_ab.txt)
    x4e_ms = float(a.get('x4e_ms', 32.7))            # teacher fp16x1 MLP kernels per frame (profiles/r05_teacher_x1_embed_ab.txt)
    body8_ms = float(a.get('body8_ms', 11.29))       # r2l_body8_kernel per 800 x 800 frame (profiles/r05_bench_n1_fp16_e4m3.json)
    print('tools/energy_account_r05.py: joules per instruction from profiles/r04_energy_account.txt (section 2), counts from the committed streams,')
    print('times from this round (bodyx %.2f ms per 800x800 frame, teacher three-pass chain %.1f ms and single-pass chain %.1f ms per 400x400 frame)' % (bodyx_ms, p3_ms, x4e_ms))
    print()
    account('1. r2l_bodyx_kernel (R2L fp16x3_asm: the rung the trained-like student lands on, max|a| 126)',
            classes(counts(os.path.join(CSRC, 'r2l_bodyx_asm.inc'), 'L_block_', 's_cbranch_scc1 L_block_')),
            bodyx_ms * 1e-3 / (20 * 43), 'ResMLP block of a 128-ray tile per wave: %.2f ms / (20 tiles per workgroup x 43 blocks)' % bodyx_ms)
    account('2. nerf_chain_kernel<false, 2, true> (teacher fp16x3_asm: the rung every trained teacher lands on)',
            classes(counts(os.path.join(CSRC, 'nerf_mlpp3_asm.inc'))), p3_ms * 1e-3 / 1250,
            '128-point tile (11 layers) per wave: %.1f ms / 1,250 tiles per CU' % p3_ms)
    account('4. r2l_body8_kernel (R2L fp16_e4m3 / fp16_split8: what the trained-like student lands on since the split rungs -- e4m3 terms behind a three-pass head)',
            classes(counts(os.path.join(CSRC, 'r2l_body8_asm.inc'), 'L_block_', 's_cbranch_scc1 L_block_')),
            body8_ms * 1e-3 / (20 * 43), 'ResMLP block of a 128-ray tile per wave: %.2f ms / (20 tiles per workgroup x 43 blocks)' % body8_ms)
    c = counts(os.path.join(CSRC, 'nerf_mlpx4e_asm.inc'), 'L_tile_', 's_cbranch_scc1 L_tile_')
    account('3. nerf_chain_emb_kernel (teacher fp16x1, embedding in the stream)', classes(c), x4e_ms * 1e-3 / 625,
            '256-point tile per wave: %.1f ms / 625 tiles per CU' % x4e_ms)


if __name__ == '__main__':
    main()
