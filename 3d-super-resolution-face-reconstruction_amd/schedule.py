"""Noise schedule on the host, in float64, cast to fp32 at the end — the numbers the HIP sampler
consumes (sr3_set_schedule).

Follows GaussianDiffusion.set_new_noise_schedule and make_beta_schedule of the reference
(model/sr/sr3_modules/diffusion.py:12-50, 93-142); formulas as summarised in SURVEY.md §8a.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

BUFFER_NAMES = (
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
    "posterior_mean_coef1", "posterior_mean_coef2",
)
# the five per-step arrays the sampler reads (diffusion.py:144-162,182-187)
ENGINE_BUFFERS = ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                  "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")


def make_beta_schedule(schedule: str, n_timestep: int, linear_start: float = 1e-4,
                       linear_end: float = 2e-2, cosine_s: float = 8e-3) -> np.ndarray:
    T = int(n_timestep)
    f8 = np.float64
    if schedule == "linear":
        return np.linspace(linear_start, linear_end, T, dtype=f8)
    if schedule == "quad":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, T, dtype=f8) ** 2
    if schedule in ("warmup10", "warmup50"):
        frac = 0.1 if schedule == "warmup10" else 0.5
        betas = np.full(T, linear_end, dtype=f8)
        k = int(T * frac)
        betas[:k] = np.linspace(linear_start, linear_end, k, dtype=f8)
        return betas
    if schedule == "const":
        return np.full(T, linear_end, dtype=f8)
    if schedule == "jsd":
        return 1.0 / np.linspace(T, 1, T, dtype=f8)
    if schedule == "cosine":
        ts = np.arange(T + 1, dtype=f8) / T + cosine_s
        ac = np.cos(ts / (1 + cosine_s) * math.pi / 2) ** 2
        ac = ac / ac[0]
        return np.minimum(1 - ac[1:] / ac[:-1], 0.999)
    raise NotImplementedError(schedule)


def schedule_buffers(schedule_opt) -> Dict[str, np.ndarray]:
    """All 12 registered fp32 buffers plus `sqrt_alphas_cumprod_prev` (float64, length T+1, the
    reference keeps it as a numpy attribute) and its fp32 cast `noise_level`."""
    betas = make_beta_schedule(schedule_opt["schedule"], schedule_opt["n_timestep"],
                               schedule_opt["linear_start"], schedule_opt["linear_end"])
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    sqrt_ac_prev = np.sqrt(np.append(1.0, ac))
    var = betas * (1.0 - ac_prev) / (1.0 - ac)
    b64 = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": np.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": np.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": np.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": np.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": np.sqrt(1.0 / ac - 1),
        "posterior_variance": var,
        "posterior_log_variance_clipped": np.log(np.maximum(var, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(ac_prev) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac),
    }
    out = {k: v.astype(np.float32) for k, v in b64.items()}
    out["sqrt_alphas_cumprod_prev"] = sqrt_ac_prev
    out["noise_level"] = sqrt_ac_prev.astype(np.float32)
    return out


def sample_inter(T: int) -> int:
    return 1 | (int(T) // 10)


def frame_steps(T: int):
    """Steps i (descending) after which the reference appends a frame (diffusion.py:192,209-211)."""
    si = sample_inter(T)
    return [i for i in reversed(range(int(T))) if i % si == 0]
