#!/usr/bin/env python
"""Writes profiles/<tag>_kernel_resources.md: register / spill / scratch numbers of every gfx950 kernel of the library,
from the metadata hipcc emits with -S (cross-compiles without a GPU).  usage: python tools/kernel_resources.py r02"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'efficient-nerf_amd', 'csrc')
tag = sys.argv[1] if len(sys.argv) > 1 else 'rXX'
rows = []
with tempfile.TemporaryDirectory() as d:
    for f in ('r2l_kernels.hip', 'r2l_body.hip', 'nerf_kernels.hip'):
        s = os.path.join(d, f + '.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '--cuda-device-only',
                        '-S', os.path.join(CSRC, f), '-o', s], check=True, stderr=subprocess.DEVNULL)
        text = open(s).read()
        meta = text[text.index('amdhsa.kernels:'):]
        for blk in re.split(r'\n  - \.agpr_count:', meta)[1:]:
            blk = '.agpr_count:' + blk
            g = lambda k: re.search(r'\.%s:\s+(\S+)' % k, blk).group(1)
            name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
            name = re.sub(r'\(.*', '', name)
            rows.append((name, g('vgpr_count'), g('agpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'),
                         g('private_segment_fixed_size'), g('group_segment_fixed_size')))
out = ['# gfx950 resource usage of every kernel (hipcc -S metadata, ROCm 7.2; tools/kernel_resources.py)', '',
       'The bench-default path (R2L_PREC_FP16_FP8) launches `r2l_head_kernel` and `r2l_body_kernel` (which ends every ray tile '
       'with the tail layer; every 8th launch its range-guard build `r2l_body_guard_kernel`); R2L_PREC_FP16_E4M3 '
       '`r2l_body8_kernel` / `r2l_body8_guard_kernel`; the teacher in fp16_fp8 `nerf_chain_kernel`.',
       'The 24 / 48 B of scratch are the by-value kernel argument block indexed dynamically (`c2w_host`), not spills.', '',
       'kernel | vgpr_count (arch+acc) | agpr | vgpr_spill | sgpr_spill | scratch bytes | static LDS',
       '---|---|---|---|---|---|---']
out += ['`%s` | %s | %s | %s | %s | %s | %s' % r for r in rows]
path = os.path.join(ROOT, 'profiles', tag + '_kernel_resources.md')
open(path, 'w').write('\n'.join(out) + '\n')
print(open(path).read())
