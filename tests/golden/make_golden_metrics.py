"""Golden vectors for the test-report metrics: the REFERENCE's utils/ssim_torch.py (ssim) and
utils/run_nerf_raybased_helpers.py (img2mse, mse2psnr) run on seeded images.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden_metrics.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('R2L_REFERENCE', '/root/reference')
sys.path.insert(0, REF)

import utils.run_nerf_raybased_helpers as RH  # noqa: E402  (reference)
from utils.ssim_torch import ssim as ref_ssim  # noqa: E402  (reference)

torch.set_grad_enabled(False)


def smooth(g, h, w):
    """A natural-image-like field: low-frequency noise upsampled + a little pixel noise, in [0, 1]."""
    base = torch.rand(1, 3, h // 8 + 2, w // 8 + 2, generator=g)
    img = torch.nn.functional.interpolate(base, size=(h, w), mode='bilinear', align_corners=True)[0]
    return (img + 0.05 * torch.randn(3, h, w, generator=g)).clamp(0, 1).permute(1, 2, 0).contiguous()


if __name__ == '__main__':
    g = torch.Generator().manual_seed(11)
    out = {}
    cases = []
    for i, (h, w) in enumerate([(48, 56), (40, 40), (17, 23)]):
        a = smooth(g, h, w)
        b = (a + (0.02 * (i + 1)) * torch.randn(h, w, 3, generator=g)).clamp(0, 1)
        cases.append((a, b))
    cases.append((cases[0][0], cases[0][0].clone()))  # identical images: ssim = 1
    cases.append((torch.ones(32, 32, 3), torch.zeros(32, 32, 3)))  # white vs black
    for i, (a, b) in enumerate(cases):
        out[f'a_{i}'], out[f'b_{i}'] = a.numpy(), b.numpy()
        # main.py:46: ssim_(unsqueeze(img.permute(2,0,1), 0), unsqueeze(ref.permute(2,0,1), 0))
        out[f'ssim_{i}'] = ref_ssim(a.permute(2, 0, 1).unsqueeze(0), b.permute(2, 0, 1).unsqueeze(0)).numpy()
        mse = RH.img2mse(a, b)
        out[f'mse_{i}'] = mse.numpy()
        out[f'psnr_{i}'] = RH.mse2psnr(mse).numpy() if float(mse) > 0 else np.array([np.inf], dtype=np.float32)
    # ndc_rays (helpers:260-279) on forward-facing and awkward rays
    ro = torch.randn(64, 3, generator=g) * 0.3
    rd = torch.randn(64, 3, generator=g)
    rd[:, 2] = -rd[:, 2].abs() - 0.2   # looking down -z like LLFF cameras
    rd[0] = torch.tensor([0., 0., -1.])
    ro[1] = torch.tensor([0., 0., 0.])
    for (H_, W_, f_) in ((378, 504, 407.5657), (400, 400, 555.5555155968841)):
        o2, d2 = RH.ndc_rays(H_, W_, f_, 1., ro, rd)
        out[f'ndc_o_{H_}'], out[f'ndc_d_{H_}'] = o2.numpy(), d2.numpy()
    out['ndc_in_o'], out['ndc_in_d'] = ro.numpy(), rd.numpy()
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **out)
    print('metrics.npz', len(out), 'arrays', os.path.getsize(os.path.join(HERE, 'metrics.npz')) // 1024, 'KiB')
    print({k: float(np.ravel(v)[0]) for k, v in out.items() if k.startswith(('ssim', 'psnr'))})
