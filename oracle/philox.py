"""CPU twin of the device RNG (csrc/kernels_misc.hip philox_normal) — TEST INFRASTRUCTURE ONLY.

Philox4x32-10 (Salmon et al., SC'11) keyed by the 64-bit seed, counter =
(element // 4, draw, image_lo, image_hi); normals by Box-Muller on 24-bit uniforms:
element e uses the pair (r[2p], r[2p+1]) with p = (e >> 1) & 1; even e -> r*cos, odd e -> r*sin.
The reference has no counterpart: it draws from torch's global generator (diffusion.py:186,205),
which cannot be reproduced across devices — parity tests inject noise instead.
"""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n1 = p1 & MASK
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0 & MASK, n1, n2 & MASK, n3
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def normal(seed: int, image: int, draw: int, n: int) -> np.ndarray:
    """n standard normals of draw `draw` for image `image` (fp32)."""
    e = np.arange(n, dtype=np.uint64)
    z = np.zeros(n, dtype=np.uint64)
    r = philox4x32_10(e >> np.uint64(2), z + np.uint64(draw), z + np.uint64(image & 0xFFFFFFFF),
                      z + np.uint64((image >> 32) & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    r = np.stack(r, axis=0)
    pair = ((e >> np.uint64(1)) & np.uint64(1)).astype(np.int64)
    idx = np.arange(n)
    a = r[2 * pair, idx]
    b = r[2 * pair + 1, idx]
    f = np.float32(5.9604644775390625e-08)
    u1 = ((a >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * f
    u2 = ((b >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * f
    rad = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    th = (np.float32(6.283185307179586) * u2).astype(np.float32)
    odd = (e & np.uint64(1)).astype(bool)
    return np.where(odd, rad * np.sin(th), rad * np.cos(th)).astype(np.float32)


def noise_slabs(seed: int, T: int, B: int, C: int, H: int, W: int, image_offset: int = 0) -> np.ndarray:
    """[T,B,C,H,W] exactly as the device draws them: draw 0 = initial image, draw k = step T-k."""
    out = np.empty((T, B, C, H, W), dtype=np.float32)
    for k in range(T):
        for b in range(B):
            out[k, b] = normal(seed, image_offset + b, k, C * H * W).reshape(C, H, W)
    return out
