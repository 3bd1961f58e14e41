#!/usr/bin/env python3
"""Operand-format study on the CPU (development script; TEST INFRASTRUCTURE like the oracle it drives —
not collected by pytest, never imported by the product).

Question: how far does a sampler drift from the reference when every conv product x·w is evaluated as a
sum of partial products of narrower operands? The GPU path computes (profiles/README.md)

    f16x3   xh·wh + xl·wh + xh·wl            x = xh + xl, w = wh + wl, all four fp16 (three f16 MFMAs)

and the candidates price the two CORRECTION terms at a cheaper MFMA rate:

    f16x2   xh·wh + xl·wh                    (weights' lo dropped; round-1 finding 22: 1.7e-3 — the control)
    bf8c    xh·wh + b8(xl)·b8(wh) + b8(xh)·b8(wl)      b8 = OCP e5m2 = the rounded top byte of the fp16 value
    fp8c    xh·wh + s8(xl)·s8(wh) + s8(xh)·s8(wl)      s8 = OCP e4m3 with one power-of-two scale per 32 channels
                                                        (the block scale of v_mfma_scale_f32_16x16x128_f8f6f4)

The partial products themselves are accumulated by torch's CPU conv in fp32 (the MFMAs accumulate in fp32
too); only the operand formats are emulated. Runs the reference-made sampler fixtures of tests/golden/ through
oracle/sr3_oracle_aten.py with `F.conv2d` replaced, and prints max-abs error of the final image and every
recorded frame against the reference's own output.

    python tests/emulate_operand_formats.py [tiny|cfg1|cfg2head|cfg2] [mode ...]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))

import sr3_oracle_aten as aten                      # noqa: E402
from conftest import cfg_from_meta, load_golden, pkg   # noqa: E402

synth = pkg("synth")


def split16(v):
    h = v.to(torch.float16).to(torch.float32)
    return h, (v - h).to(torch.float16).to(torch.float32)


def b8(v):
    return v.to(torch.float8_e5m2).to(torch.float32)


def s8(v, dim):
    """e4m3 with one power-of-two scale per block of 32 along `dim` (block maximum scaled into [128, 256))."""
    v = v.movedim(dim, -1)
    n = v.shape[-1]
    pad = (-n) % 32
    vp = F.pad(v, (0, pad)).reshape(*v.shape[:-1], (n + pad) // 32, 32)
    mx = vp.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(torch.clamp(mx, min=2.0 ** -120)))
    sc = torch.exp2(7.0 - e)
    q = (vp * sc).to(torch.float8_e4m3fn).to(torch.float32) / sc
    return q.reshape(*v.shape[:-1], n + pad)[..., :n].movedim(-1, dim)


SAT = {"n": 0, "tot": 0, "max": 0.0}


def s8_fixed(v, log2scale):
    """e4m3 with ONE fixed power-of-two scale (no per-block scale to store); values beyond +-448 saturate and are counted."""
    sv = v * (2.0 ** log2scale)
    SAT["n"] += int((sv.abs() > 448).sum())
    SAT["tot"] += sv.numel()
    SAT["max"] = max(SAT["max"], float(v.abs().max()))
    return sv.clamp(-448, 448).to(torch.float8_e4m3fn).to(torch.float32) * (2.0 ** -log2scale)


def s8_col(w):
    """e4m3 with one power-of-two scale per output channel (foldable into the existing per-column pre-scale)."""
    mx = w.abs().amax(dim=(1, 2, 3), keepdim=True)
    e = torch.floor(torch.log2(torch.clamp(mx, min=2.0 ** -120)))
    sc = torch.exp2(7.0 - e)
    return (w * sc).to(torch.float8_e4m3fn).to(torch.float32) / sc


class ConvProxy:
    """Stands in for torch.nn.functional inside the oracle module: conv2d with emulated operand formats."""

    def __init__(self, mode):
        self.mode = mode

    def __getattr__(self, name):
        return getattr(F, name)

    def conv2d(self, x, w, b=None, stride=1, padding=0):
        m = self.mode
        if m == "f32":
            return F.conv2d(x, w, b, stride=stride, padding=padding)
        xh, xl = split16(x)
        wh, wl = split16(w)
        y = F.conv2d(xh, wh, None, stride=stride, padding=padding)
        if m == "f16x3":
            y = y + F.conv2d(xl, wh, None, stride=stride, padding=padding) + F.conv2d(xh, wl, None, stride=stride, padding=padding)
        elif m == "f16x2":
            y = y + F.conv2d(xl, wh, None, stride=stride, padding=padding)
        elif m == "bf8c":
            y = y + F.conv2d(b8(xl), b8(wh), None, stride=stride, padding=padding) + \
                F.conv2d(b8(xh), b8(wl), None, stride=stride, padding=padding)
        elif m == "fp8c":
            y = y + F.conv2d(s8(xl, 1), s8(wh, 1), None, stride=stride, padding=padding) + \
                F.conv2d(s8(xh, 1), s8(wl, 1), None, stride=stride, padding=padding)
        elif m.startswith("fp8f"):
            # fixed activation scales: xh * 2^A, xl * 2^(A + 11); per-output-channel weight scales
            A = int(m[4:] or 4)
            y = y + F.conv2d(s8_fixed(xl, A + 11), s8_col(wh), None, stride=stride, padding=padding) + \
                F.conv2d(s8_fixed(xh, A), s8_col(wl), None, stride=stride, padding=padding)
        else:
            raise ValueError(m)
        return y if b is None else y + b.reshape(1, -1, 1, 1)


def run(fixture, mode, max_steps=None):
    g = load_golden(fixture)
    m = g["meta"]
    cfg = cfg_from_meta(m)
    sd = synth.synth_state_dict(cfg, m["seed"])
    sch = aten.noise_schedule(m["schedule"])
    B, r, T = m["B"], m["r"], m["schedule"]["n_timestep"]
    noise = synth.synth_noise(T, B, 3, r, r, m["seed"])
    cond = g["cond"] if m.get("conditional", True) else None
    aten.F = ConvProxy(mode)
    t0 = time.time()
    try:
        if "frames_sub" in g:                      # the T = 1000 fixture: strided frames + the final image
            st, si = m["frame_stride"], 1 | (T // 10)
            tsd = aten.to_torch_state(sd)
            tc = torch.from_numpy(cond)
            x = torch.from_numpy(noise[0].copy())
            errs, nf = [], 0
            with torch.no_grad():
                for k, t in enumerate(reversed(range(T))):
                    x = aten.p_sample(tsd, cfg, sch, x, t, tc, torch.from_numpy(noise[k + 1].copy()) if t > 0 else None)
                    if t % si == 0:
                        errs.append(float(np.abs(x.numpy()[..., ::st, ::st] - g["frames_sub"][nf]).max()))
                        nf += 1
                        print(f"    [{mode}] t={t} frame {nf - 1}: {errs[-1]:.3e}  ({time.time() - t0:.0f} s)", flush=True)
                    if max_steps is not None and k + 1 >= max_steps and t % si == 0:
                        break
            fin = float(np.abs(x.numpy() - g["final"]).max()) if nf == 10 else float("nan")
            return fin, errs
        final, frames = aten.p_sample_loop(sd, cfg, sch, cond, noise)
        first = cond if cond is not None else noise[0]
        ret = np.concatenate([first, frames.reshape(-1, *frames.shape[2:])], axis=0)
        nfr = frames.shape[0]
        per = np.abs(ret - g["ret_img"]).reshape(nfr + 1, -1).max(axis=1) if ret.shape[0] == nfr + 1 else \
            np.abs(ret - g["ret_img"]).reshape(ret.shape[0], -1).max(axis=1)
        return float(np.abs(final[-1] - g["last"]).max()), [float(v) for v in per]
    finally:
        aten.F = F


FIX = {"tiny": ("sampler_tiny.npz", None), "cfg1": ("sampler_cfg1_8_16.npz", None),
       "cfg5": ("sampler_cfg5_32_128.npz", None),
       "cfg2head": ("sampler_cfg2_16_128_T1000.npz", 91), "cfg2": ("sampler_cfg2_16_128_T1000.npz", None)}

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    modes = sys.argv[2:] or ["f32", "f16x3", "f16x2", "bf8c", "fp8c"]
    torch.set_num_threads(int(os.environ.get("EMU_THREADS", "8")))
    fixture, max_steps = FIX[which]
    for mode in modes:
        t0 = time.time()
        fin, per = run(fixture, mode, max_steps)
        extra = f"  saturated {SAT['n']}/{SAT['tot']} max|x| {SAT['max']:.1f}" if mode.startswith("fp8f") else ""
        print(f"{which:9s} {mode:6s} final {fin:.3e}  worst frame {max(per):.3e}  ({time.time() - t0:.0f} s){extra}", flush=True)
        SAT.update(n=0, tot=0, max=0.0)
