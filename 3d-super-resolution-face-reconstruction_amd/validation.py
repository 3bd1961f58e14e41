"""Batched multi-sample validation: what the reference does one image and
one sample at a time (`for val_data: for k in range(cfg.sample): model.test_val(...)`,
lib/trainer_temp.py:441-446 -> model/sr3d/model.py:368-375,428-433) as ONE sharded batch of
images x samples through the HIP sampler, followed by the PSNR / SSIM of core/metrics.py.

SSIM restates core/metrics.py:84-104 with numpy only: `cv2.filter2D(img, -1, window)[5:-5, 5:-5]`
with an 11x11 window is exactly the 'valid' correlation with that window, and
`cv2.getGaussianKernel(11, 1.5)` is the normalised exp(-(i-5)^2 / (2*1.5^2)). cv2 is not available
in the build container, so this file's SSIM is checked against an independent scipy evaluation,
not against cv2 itself (parity unpinned for the cv2 call, formula-level faithful).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

from .metrics import psnr, tensor2img


def gaussian_kernel(ksize: int = 11, sigma: float = 1.5) -> np.ndarray:
    i = np.arange(ksize, dtype=np.float64) - (ksize - 1) / 2.0
    k = np.exp(-(i * i) / (2.0 * sigma * sigma))
    return k / k.sum()


def _valid_filter(img: np.ndarray, k1: np.ndarray) -> np.ndarray:
    # separable 'valid' correlation (the window is an outer product of a symmetric kernel)
    n = k1.size
    h = sum(k1[j] * img[:, j:img.shape[1] - n + 1 + j] for j in range(n))
    return sum(k1[j] * h[j:h.shape[0] - n + 1 + j, :] for j in range(n))


def ssim(img1: np.ndarray, img2: np.ndarray) -> float:
    """core/metrics.py:84-104 for one 2-D image pair in [0, 255]."""
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    a, b = img1.astype(np.float64), img2.astype(np.float64)
    k = gaussian_kernel(11, 1.5)
    mu1, mu2 = _valid_filter(a, k), _valid_filter(b, k)
    mu1_sq, mu2_sq, mu1_mu2 = mu1 ** 2, mu2 ** 2, mu1 * mu2
    s1 = _valid_filter(a * a, k) - mu1_sq
    s2 = _valid_filter(b * b, k) - mu2_sq
    s12 = _valid_filter(a * b, k) - mu1_mu2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return float(m.mean())


def calculate_ssim(img1: np.ndarray, img2: np.ndarray) -> float:
    """core/metrics.py:107-125. For HxWx3 input the reference averages three evaluations of
    `ssim(img1, img2)` on the WHOLE 3-channel arrays (it never indexes the channel); filter2D treats
    channels independently, so that equals the mean of the per-channel SSIM maps — computed so here."""
    if img1.shape != img2.shape:
        raise ValueError("Input images must have the same dimensions.")
    if img1.ndim == 2:
        return ssim(img1, img2)
    if img1.ndim == 3:
        if img1.shape[2] == 3:
            return float(np.mean([ssim(img1[..., c], img2[..., c]) for c in range(3)]))
        if img1.shape[2] == 1:
            return ssim(np.squeeze(img1), np.squeeze(img2))
    raise ValueError("Wrong input image dimensions.")


def validate_batch(netG, sr: "torch.Tensor", hr: "torch.Tensor", samples: int = 1,
                   seed: Optional[int] = None, sharded: bool = False) -> Dict[str, np.ndarray]:
    """Runs `samples` independent SR3 chains per conditioning image as one batch of N*samples
    images (sample k of image i is batch row k*N + i) and scores them against `hr`.

    sr, hr: [N,3,r,r] in [-1,1] (the dataset's 'SR' and 'HR' entries, datasets/LRHR_dataset.py:93-99).
    N * samples may exceed what one library call takes (~250 images at 128x128): `sample_batch` runs equal chunks
    with the Philox streams keyed by the global row, so the scores do not depend on the chunking.
    Returns per (sample, image) PSNR / SSIM arrays and their means. The reference's running
    `avg / idx * sample` (lib/trainer_temp.py:445-446) multiplies by `sample` instead of dividing —
    the plain means are reported here.
    """
    import torch
    from . import dist as _dist

    N = sr.shape[0]
    x = sr.repeat(samples, 1, 1, 1)
    if seed is None:
        seed = netG._draw_seed()
        if sharded and torch.distributed.is_available() and torch.distributed.is_initialized():
            # every rank must key the Philox streams with the SAME seed (image i's stream does not
            # depend on the world size, dist.py): rank 0's draw wins
            box = [seed]
            torch.distributed.broadcast_object_list(box, src=0)
            seed = int(box[0])
    if sharded:
        out = _dist.sharded_super_resolution(
            lambda xs, off: netG.super_resolution_batch(xs, seed=seed, image_offset=off), x)
    else:
        out = netG.super_resolution_batch(x, seed=seed)
    out_np, hr_np = out.float().cpu().numpy(), hr.float().cpu().numpy()
    ps = np.zeros((samples, N)); ss = np.zeros((samples, N))
    for k in range(samples):
        for i in range(N):
            a, b = tensor2img(out_np[k * N + i]), tensor2img(hr_np[i])
            ps[k, i] = psnr(a, b)
            ss[k, i] = calculate_ssim(a, b)
    finite = np.isfinite(ps)
    return {"psnr": ps, "ssim": ss, "images": out,
            "mean_psnr": float(ps[finite].mean()) if finite.any() else math.inf,
            "mean_ssim": float(ss.mean())}
