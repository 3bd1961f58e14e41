"""Deterministic synthetic inputs (weights, conditioning images, noise) shared by the golden
fixture generator, the tests and bench.py. numpy's legacy RandomState streams are frozen, so the
same tensors are regenerated bit-identically here and on the GPU box without shipping them.
No pretrained SR3 checkpoint is available offline."""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

from .graph import UNetConfig, param_specs


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0xFFFFFFFF)


def synth_state_dict(cfg: UNetConfig, seed: int = 0, prefix: str = "") -> Dict[str, np.ndarray]:
    """Weights ~ N(0, 1/fan_in), GroupNorm gamma = 1 + 0.1 N, every bias / beta = 0.1 N
    (SURVEY.md §8d), keyed per parameter name so the order of generation does not matter."""
    sd = {}
    for name, shape, kind in param_specs(cfg):
        rs = _rs(name, seed)
        if kind in ("conv", "linear"):
            fan_in = int(np.prod(shape[1:]))
            a = rs.standard_normal(shape) / np.sqrt(fan_in)
        elif kind == "norm_w":
            a = 1.0 + 0.1 * rs.standard_normal(shape)
        else:
            a = 0.1 * rs.standard_normal(shape)
        sd[prefix + name] = a.astype(np.float32)
    return sd


def synth_cond(B: int, r: int, l: int, seed: int = 0) -> np.ndarray:
    """Conditioning batch [B,3,r,r] in [-1,1]: a random l x l image upsampled (cubic spline) to
    r x r, the shape of the reference's bicubic `SR` input (datasets/tool/prepare_data.py:37-47)."""
    from scipy.ndimage import zoom
    lo = np.random.RandomState(seed ^ 0x5EED).uniform(-1, 1, (B, 3, l, l))
    up = zoom(lo, (1, 1, r / l, r / l), order=3, mode="nearest")
    return np.clip(up, -1, 1).astype(np.float32)


def synth_noise(T: int, B: int, C: int, H: int, W: int, seed: int = 0) -> np.ndarray:
    """Noise slabs [T,B,C,H,W]: slab 0 is the initial image (torch.randn in diffusion.py:205),
    slab k the randn_like of loop iteration k-1, i.e. of step t = T-k (diffusion.py:186)."""
    return np.random.RandomState(seed ^ 0xA015E).standard_normal((T, B, C, H, W)).astype(np.float32)


# the reference's yml UNet (config/*.yml:35-48,60-63); image_size 224 is what every yml says,
# 128 is BASELINE.json's "attention-heavy" reading
def yml_unet_config(image_size: int = 224) -> UNetConfig:
    return UNetConfig(in_channel=6, out_channel=3, inner_channel=64, norm_groups=32,
                      channel_mults=(1, 2, 4, 8, 8), attn_res=(16,), res_blocks=2, dropout=0.2,
                      image_size=image_size)


def tiny_unet_config() -> UNetConfig:
    return UNetConfig(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                      channel_mults=(1, 2), attn_res=(8,), res_blocks=1, dropout=0.0, image_size=16)


def yml_opt(l: int, r: int, n_timestep: int, image_size: int = 224, phase: str = "val") -> dict:
    """Plain-dict equivalent of config/sr_sr3_VGGF2_<l>_<r>_model*.yml restricted to the keys
    define_G and set_new_noise_schedule read."""
    sched = {"schedule": "linear", "n_timestep": n_timestep, "linear_start": 1e-6, "linear_end": 1e-2}
    c = yml_unet_config(image_size)
    return {
        "phase": phase,
        "sr": {"model": {
            "which_model_G": "sr3",
            "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": c.inner_channel,
                     "channel_multiplier": list(c.channel_mults), "attn_res": list(c.attn_res),
                     "res_blocks": c.res_blocks, "dropout": c.dropout},
            "beta_schedule": {"train": dict(sched), "val": dict(sched)},
            "diffusion": {"image_size": image_size, "channels": 3, "conditional": True},
        }},
    }
