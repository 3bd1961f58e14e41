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


def batch_psnr_stats(a_nchw: np.ndarray, b_nchw: np.ndarray) -> dict:
    """Per-batch PSNR summary of two batches of [-1,1] images: `mean_db` over the image pairs that
    differ after uint8 rounding (None when there is none), `identical` = number of pairs that are
    identical after rounding (their PSNR is infinite and is NOT part of the mean), `n` = pairs."""
    vals = [psnr(tensor2img(x), tensor2img(y)) for x, y in zip(a_nchw, b_nchw)]
    finite = [v for v in vals if math.isfinite(v)]
    return {"mean_db": float(np.mean(finite)) if finite else None,
            "identical": len(vals) - len(finite), "n": len(vals)}


def batch_psnr(a_nchw: np.ndarray, b_nchw: np.ndarray) -> float:
    """Mean PSNR over the pairs that differ after rounding; inf if every pair is identical.
    Pairs with infinite PSNR are excluded from the mean — use batch_psnr_stats to see how many."""
    st = batch_psnr_stats(a_nchw, b_nchw)
    return float("inf") if st["mean_db"] is None else st["mean_db"]
