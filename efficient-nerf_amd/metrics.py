"""Image metrics of the reference's test report (`[TEST] TestPSNR .. TestSSIM ..`,
main.py:331-335, 384-391): PSNR from the MSE (helpers:18-20) and the Gaussian-window SSIM of
utils/ssim_torch.py.  Plain torch on whatever device the images live on: reporting, not hot path."""
import math

import torch
import torch.nn.functional as F


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(mse):
    """-10 * log(mse) / log(10)  (utils/run_nerf_raybased_helpers.py:18-20)."""
    mse = torch.as_tensor(mse)
    return -10. * torch.log(mse) / torch.log(torch.tensor([10.], device=mse.device))


def ssim_window(window_size, channel, sigma=1.5):
    """utils/ssim_torch.py:11-25: normalised 1-D Gaussian, outer product, one copy per channel."""
    g = torch.Tensor([math.exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2 = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/ssim_torch.py:28-53, 86-94.  img1, img2: [N, C, H, W]."""
    channel = img1.shape[1]
    window = ssim_window(window_size, channel).to(img1.device).type_as(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def ssim_hwc(img, ref):
    """main.py:46: ssim of two [H, W, 3] images (the reference permutes to [1, C, H, W])."""
    return ssim(img.permute(2, 0, 1).unsqueeze(0), ref.permute(2, 0, 1).unsqueeze(0))
