"""GPU: the compiler-scheduled kernels stay exact while another process time-shares the card.

The multi-rank rehearsals of this suite run several processes on ONE GPU (the deployment is one process per GPU).  Round 4 found
that beside another process's nerf_chain_kernel a build of nerf_get_rays_kernel with packed-fp32 VALU ops (formed by the SLP
vectorizer) returned wrong d.x in groups of 16 lanes, which made the two-rank `create_data` directory differ from the one-rank one
at the reference's sizes; the library is therefore built with -fno-slp-vectorize (csrc/Makefile).  This pins it: get_rays
(utils/run_nerf_raybased_helpers.py:231-257) must equal the CPU oracle bit for bit in that situation."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_get_rays_is_exact_beside_another_process_chain_kernel():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gpu_sharing_check.py'), 'c'], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('mode c')]
    assert r.returncode == 0 and line and 'oracle 0;' in line[0] and 'CPU 0;' in line[0] and 'first render 0;' in line[0] and line[0].rstrip().endswith('(generated head + body) 0'), r.stdout[-1500:] + r.stderr[-1500:]
