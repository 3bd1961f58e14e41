"""Image-quality half of the headline metric: uint8 conversion and PSNR as the reference defines
them (core/metrics.py:16-42 tensor2img for a single CHW image, :74-81 calculate_psnr)."""
from __future__ import annotations

import math

import numpy as np


def tensor2img(chw: np.ndarray, min_max=(-1.0, 1.0)) -> np.ndarray:
    """[C,H,W] float in any range -> [H,W,C] uint8: clamp, map to [0,1], x255, round."""
    a = np.clip(np.asarray(chw, dtype=np.float32), *min_max)
    a = (a - min_max[0]) / (min_max[1] - min_max[0])
    return np.round(np.transpose(a, (1, 2, 0)) * np.float32(255.0)).astype(np.uint8)


def psnr(img1: np.ndarray, img2: np.ndarray) -> float:
    d = img1.astype(np.float64) - img2.astype(np.float64)
    mse = float(np.mean(d * d))
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def batch_psnr(a_nchw: np.ndarray, b_nchw: np.ndarray) -> float:
    """Mean PSNR over a batch of [-1,1] images (inf if every image is identical after rounding)."""
    vals = [psnr(tensor2img(x), tensor2img(y)) for x, y in zip(a_nchw, b_nchw)]
    finite = [v for v in vals if math.isfinite(v)]
    if not finite:
        return float("inf")
    return float(np.mean(finite)) if len(finite) == len(vals) else float(np.mean(finite))
