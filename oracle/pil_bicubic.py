"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.  8-bit bicubic resize exactly as Pillow does it.

The reference builds its conditioning image with PIL: `resize_multiple` in
/root/reference/datasets/tool/prepare_data.py:24-47 (torchvision `resize` on a PIL image ==
`Image.resize(size, Image.BICUBIC)`), then `ToTensor` and `x*2-1`
(/root/reference/datasets/util.py:76-83). Pillow is a third-party dependency (12.2.0 in the build
container); its published algorithm (src/libImaging/Resample.c: precompute_coeffs,
normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc) is restated here and
**pinned** against Pillow itself by tests/golden/preproc_bicubic.npz (made by
tests/golden/make_golden_preproc.py).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (bounds [out,2] int32 (xmin, count), kk [out, ksize] int32 fixed point, ksize)."""
    support0 = 2.0
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = support0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def _clip8(v: np.ndarray) -> np.ndarray:
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """img [H,W,C] uint8 -> [out_h,out_w,C] uint8: horizontal pass, then vertical pass, each
    rounded to 8 bits (the two-pass scheme of ImagingResample)."""
    H, W, C = img.shape
    cur = img
    if out_w != W:
        b, kk, _ = precompute_coeffs(W, out_w)
        out = np.empty((H, out_w, C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for xx in range(out_w):
            x0, n = b[xx]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(src[:, x0:x0 + n, :], kk[xx, :n].astype(np.int64), axes=([1], [0]))
            out[:, xx, :] = _clip8(acc)
        cur = out
    if out_h != H:
        b, kk, _ = precompute_coeffs(H, out_h)
        out = np.empty((out_h, cur.shape[1], C), dtype=np.uint8)
        src = cur.astype(np.int64)
        for yy in range(out_h):
            y0, n = b[yy]
            acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[yy, :n].astype(np.int64), src[y0:y0 + n, :, :], axes=([0], [0]))
            out[yy] = _clip8(acc)
        cur = out
    return cur


def to_tensor_pm1(img_u8_hwc: np.ndarray) -> np.ndarray:
    """ToTensor + min_max (-1, 1): float32 CHW = (u8 / 255) * 2 - 1 (datasets/util.py:76-83)."""
    x = img_u8_hwc.astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(np.transpose(x * np.float32(2.0) + np.float32(-1.0), (2, 0, 1)))
