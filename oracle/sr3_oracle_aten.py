"""CPU ORACLE, aten variant — TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

The same restatement as oracle/sr3_oracle.py (same citations, same module plan, same sampler rules) with
the tensor work done by torch's CPU operators — `F.conv2d`, `F.group_norm`, `torch.einsum`, `F.linear` —
i.e. the operator library the reference's own CPU path runs on (`nn.Conv2d` / `nn.GroupNorm`,
model/sr/sr3_modules/unet.py:62,84,87). Own code: nothing is imported from /root/reference.

Why it exists: bench.py's `cpu_baseline` leg. The numpy oracle is a matmul port (nine shifted GEMMs per conv)
and runs ~8x slower than the reference's torch-CPU path; timing THIS variant on the GPU box's host cores is the
faithful stand-in for "the reference's CPU path on the same host" (`"kind": "port-aten"`).

Parity status: PINNED — tests/test_oracle_golden.py checks it against the same reference-made fixtures as the
numpy oracle (UNet forward goldens, sampler goldens incl. BASELINE config 1 and the T = 1000 head).
All citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

import sr3_oracle as _np_oracle       # schedule + module plan are shared (pure numpy / pure Python)

noise_schedule = _np_oracle.noise_schedule
unet_plan = _np_oracle.unet_plan


def to_torch_state(sd: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}


def swish(x):                                                                    # unet.py:53-55
    return x * torch.sigmoid(x)


def positional_encoding(noise_level, dim):                                       # unet.py:18-31
    count = dim // 2
    step = torch.arange(count, dtype=noise_level.dtype) / count
    enc = noise_level.reshape(-1, 1) * torch.exp(-math.log(1e4) * step.unsqueeze(0))
    return torch.cat([torch.sin(enc), torch.cos(enc)], dim=-1)


def block(sd, p, x, groups):                                                     # Block, unet.py:80-91 (eval)
    h = swish(F.group_norm(x, groups, sd[p + ".block.0.weight"], sd[p + ".block.0.bias"], eps=1e-5))
    return F.conv2d(h, sd[p + ".block.3.weight"], sd[p + ".block.3.bias"], padding=1)


def resnet_block(sd, p, x, temb, groups):                                        # unet.py:94-110, :34-50
    h = block(sd, p + ".block1", x, groups)
    nb = F.linear(temb, sd[p + ".noise_func.noise_func.0.weight"], sd[p + ".noise_func.noise_func.0.bias"])
    h = h + nb[:, :, None, None]
    h = block(sd, p + ".block2", h, groups)
    if (p + ".res_conv.weight") in sd:
        x = F.conv2d(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


def self_attention(sd, p, x, groups):                                            # unet.py:113-142, n_head = 1
    B, C, H, W = x.shape
    n = F.group_norm(x, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    # token form (one head): S[p, p'] = sum_c q[c, p] k[c, p'] / sqrt(C); P = softmax over p'; o[c, p] = sum_p' P[p, p'] v[c, p']
    qkv = F.conv2d(n, sd[p + ".qkv.weight"]).reshape(B, 3 * C, H * W)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    s = torch.bmm(q.transpose(1, 2), k) / math.sqrt(C)
    pr = torch.softmax(s, dim=-1)
    o = torch.bmm(v, pr.transpose(1, 2)).reshape(B, C, H, W)
    return F.conv2d(o, sd[p + ".out.weight"], sd[p + ".out.bias"]) + x


def unet_forward(sd: Dict[str, torch.Tensor], cfg, x, noise_level, prefix: str = "",
                 taps: Optional[dict] = None) -> torch.Tensor:
    """UNet.forward, unet.py:235-265; NCHW torch tensors, sd from to_torch_state()."""
    g = cfg.norm_groups
    t = positional_encoding(noise_level.reshape(-1).to(torch.float32), cfg.inner_channel)
    t = swish(F.linear(t, sd[prefix + "noise_level_mlp.1.weight"], sd[prefix + "noise_level_mlp.1.bias"]))
    temb = F.linear(t, sd[prefix + "noise_level_mlp.3.weight"], sd[prefix + "noise_level_mlp.3.bias"])
    downs, mid, ups = unet_plan(cfg)

    def res(p, x, attn):
        x = resnet_block(sd, prefix + p + ".res_block", x, temb, g)
        return self_attention(sd, prefix + p + ".attn", x, g) if attn else x

    feats = []
    for kind, p, attn in downs:
        if kind == "conv":
            x = F.conv2d(x, sd[prefix + p + ".weight"], sd[prefix + p + ".bias"], padding=1)
        elif kind == "down":                                                               # unet.py:68-74
            x = F.conv2d(x, sd[prefix + p + ".conv.weight"], sd[prefix + p + ".conv.bias"], stride=2, padding=1)
        else:
            x = res(p, x, attn)
        feats.append(x)
        if taps is not None:
            taps[p] = x.numpy().copy()
    for kind, p, attn in mid:
        x = res(p, x, attn)
        if taps is not None:
            taps[p] = x.numpy().copy()
    for kind, p, attn in ups:
        if kind == "up":                                                                   # unet.py:58-65
            x = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), sd[prefix + p + ".conv.weight"],
                         sd[prefix + p + ".conv.bias"], padding=1)
        else:
            x = res(p, torch.cat((x, feats.pop()), dim=1), attn)                           # unet.py:261
        if taps is not None:
            taps[p] = x.numpy().copy()
    return block(sd, prefix + "final_conv", x, g)


def p_sample(sd, cfg, sched, x, t, cond, noise, prefix=""):
    """p_mean_variance + p_sample, diffusion.py:164-187 (clip_denoised=True); torch tensors in and out."""
    B = x.shape[0]
    nl = torch.full((B, 1), float(np.float32(sched["sqrt_alphas_cumprod_prev"][t + 1])), dtype=torch.float32)   # :166-167
    inp = torch.cat([cond, x], dim=1) if cond is not None else x                                                # :170
    eps = unet_forward(sd, cfg, inp, nl, prefix)
    x0 = float(sched["sqrt_recip_alphas_cumprod"][t]) * x - float(sched["sqrt_recipm1_alphas_cumprod"][t]) * eps  # :150-151
    x0.clamp_(-1.0, 1.0)                                                                                        # :175-176
    mean = float(sched["posterior_mean_coef1"][t]) * x0 + float(sched["posterior_mean_coef2"][t]) * x           # :159-160
    if t > 0:                                                                                                   # :186-187
        return mean + noise * math.exp(0.5 * float(sched["posterior_log_variance_clipped"][t]))
    return mean


def p_sample_loop(sd, cfg, sched, cond, noise, prefix=""):
    """p_sample_loop, diffusion.py:189-215 with injected noise [T,B,C,H,W] (numpy in, numpy out; same
    conventions as sr3_oracle.p_sample_loop)."""
    T = int(sched["betas"].shape[0])
    si = 1 | (T // 10)
    tsd = to_torch_state(sd)
    tc = torch.from_numpy(cond) if cond is not None else None
    img = torch.from_numpy(np.asarray(noise[0], dtype=np.float32).copy())
    frames = []
    with torch.no_grad():
        for k, i in enumerate(reversed(range(T))):
            nz = torch.from_numpy(noise[k + 1].copy()) if i > 0 else None
            img = p_sample(tsd, cfg, sched, img, i, tc, nz, prefix)
            if i % si == 0:
                frames.append(img.numpy().copy())
    return img.numpy(), np.stack(frames, axis=0)
