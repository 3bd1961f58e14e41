#!/usr/bin/env python3
"""Generates tests/golden/*.npz by importing the REFERENCE (read-only at /root/reference) in the
build container. Run from the repo root:   python tests/golden/make_golden.py

The reference never travels to the GPU box; only these small input/output vectors do. Weights are
not stored: they are regenerated from `synth.synth_state_dict(cfg, seed)` (frozen numpy
RandomState streams). The reference's RNG calls (torch.randn / torch.randn_like,
diffusion.py:186,205) are replaced by slabs of `synth.synth_noise` so the sampler is deterministic.
"""
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
sys.path.insert(0, REF)

synth = importlib.import_module("3d-super-resolution-face-reconstruction_amd.synth")
graph = importlib.import_module("3d-super-resolution-face-reconstruction_amd.graph")

from model.sr import networks as ref_networks            # noqa: E402  (reference)
from model.sr.sr3_modules import diffusion as ref_diff   # noqa: E402  (reference)

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def opt_from_cfg(cfg, sched, conditional=True):
    return {"phase": "val", "sr": {"model": {
        "which_model_G": "sr3",
        "unet": {"in_channel": cfg.in_channel, "out_channel": cfg.out_channel,
                 "inner_channel": cfg.inner_channel, "channel_multiplier": list(cfg.channel_mults),
                 "attn_res": list(cfg.attn_res), "res_blocks": cfg.res_blocks, "dropout": cfg.dropout},
        "beta_schedule": {"train": dict(sched), "val": dict(sched)},
        "diffusion": {"image_size": cfg.image_size, "channels": 3, "conditional": conditional}}}}


def build_ref(cfg, sched, seed, conditional=True):
    netG = ref_networks.define_G(opt_from_cfg(cfg, sched, conditional))
    sd = synth.synth_state_dict(cfg, seed, prefix="denoise_fn.")
    res = netG.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not res.missing_keys, res.missing_keys
    netG.set_new_noise_schedule(sched, ["cpu"])   # list form: diffusion.py:94-95 subscripts it
    netG.eval()
    return netG


class NoiseFeed:
    """Stands in for torch.randn / torch.randn_like inside the reference sampler."""

    def __init__(self, slabs):
        self.slabs, self.k = slabs, 0

    def _next(self):
        s = torch.from_numpy(self.slabs[self.k].copy())
        self.k += 1
        return s

    def __enter__(self):
        self._r, self._rl = torch.randn, torch.randn_like
        torch.randn = lambda *a, **k: self._next()
        torch.randn_like = lambda *a, **k: self._next()
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._r, self._rl


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB")


def meta(cfg, **kw):
    d = dict(in_channel=cfg.in_channel, out_channel=cfg.out_channel, inner_channel=cfg.inner_channel,
             norm_groups=cfg.norm_groups, channel_mults=list(cfg.channel_mults),
             attn_res=list(cfg.attn_res), res_blocks=cfg.res_blocks, dropout=cfg.dropout,
             image_size=cfg.image_size)
    d.update(kw)
    return np.array(json.dumps(d))


def gen_schedules():
    cases = [("linear", 100, 1e-6, 1e-2), ("linear", 1000, 1e-6, 1e-2), ("linear", 2000, 1e-4, 2e-2),
             ("quad", 50, 1e-4, 2e-2), ("warmup10", 40, 1e-4, 2e-2), ("warmup50", 40, 1e-4, 2e-2),
             ("const", 10, 1e-4, 2e-2), ("jsd", 20, 1e-4, 2e-2), ("cosine", 30, 1e-4, 2e-2)]
    out = {"cases": np.array(json.dumps(cases))}
    tiny = synth.tiny_unet_config()
    netG = ref_networks.define_G(opt_from_cfg(tiny, {"schedule": "linear", "n_timestep": 10,
                                                     "linear_start": 1e-4, "linear_end": 2e-2}))
    for i, (s, T, a, b) in enumerate(cases):
        netG.set_new_noise_schedule({"schedule": s, "n_timestep": T, "linear_start": a, "linear_end": b}, ["cpu"])
        for k, v in netG.state_dict().items():
            if not k.startswith("denoise_fn."):
                out[f"c{i}.{k}"] = v.numpy().copy()
        out[f"c{i}.sqrt_alphas_cumprod_prev"] = np.asarray(netG.sqrt_alphas_cumprod_prev, dtype=np.float64)
    save("schedules.npz", **out)


def gen_unet(name, cfg, B, r, seed, with_taps=False):
    t0 = time.time()
    sched = {"schedule": "linear", "n_timestep": 10, "linear_start": 1e-4, "linear_end": 2e-2}
    netG = build_ref(cfg, sched, seed)
    rs = np.random.RandomState(seed + 77)
    x = rs.standard_normal((B, cfg.in_channel, r, r)).astype(np.float32)
    nl = rs.uniform(0.05, 1.0, (B, 1)).astype(np.float32)
    arrs = {"x": x, "noise_level": nl}
    unet = netG.denoise_fn
    if with_taps:
        hooks = []
        for grp in ("downs", "mid", "ups"):
            for i, m in enumerate(getattr(unet, grp)):
                hooks.append(m.register_forward_hook(
                    lambda mod, inp, out, key=f"{grp}.{i}": arrs.__setitem__("tap." + key, out.numpy().copy())))
    eps = unet(torch.from_numpy(x), torch.from_numpy(nl)).numpy()
    arrs["eps"] = eps
    keys = [(k, list(v.shape)) for k, v in unet.state_dict().items()]
    arrs["meta"] = meta(cfg, B=B, r=r, seed=seed, n_params=sum(int(np.prod(s)) for _, s in keys))
    arrs["state_dict_keys"] = np.array(json.dumps(keys))
    save(name, **arrs)
    print(f"    ({time.time() - t0:.1f}s, eps std {eps.std():.3f})")


def gen_sampler(name, cfg, sched, B, r, l, seed, conditional=True, frame_stride=1, with_last=True):
    """frame_stride > 1 (large fixtures): the intermediate frames are stored sub-sampled
    ([..., ::s, ::s]) as `frames_sub`; the final images (`final`, every pixel) and `last` stay whole.
    with_last=False (the T = 1000 fixture: one reference run is ~5 min here) skips the second,
    `continous=False` run of the reference; no `last` array is stored then."""
    t0 = time.time()
    netG = build_ref(cfg, sched, seed, conditional)
    T = sched["n_timestep"]
    noise = synth.synth_noise(T, B, 3, r, r, seed)
    arrs = {}
    if conditional:
        cond = synth.synth_cond(B, r, l, seed)
        arrs["cond"] = cond
        with NoiseFeed(noise) as nf:
            ret = netG.super_resolution(torch.from_numpy(cond), continous=True).numpy()
            assert nf.k == T, nf.k
        last = None
        if with_last:
            with NoiseFeed(noise):
                last = netG.super_resolution(torch.from_numpy(cond), continous=False).numpy()
    else:
        with NoiseFeed(noise) as nf:
            ret = netG.sample(batch_size=B, continous=True).numpy()
            assert nf.k == T, nf.k
        with NoiseFeed(noise):
            last = netG.sample(batch_size=B, continous=False).numpy()
    if frame_stride > 1:
        nf = ret.shape[0] // B - 1
        fr = ret[B:].reshape(nf, B, 3, r, r)
        arrs.update(frames_sub=fr[..., ::frame_stride, ::frame_stride].copy(), final=fr[-1].copy(),
                    meta=meta(cfg, B=B, r=r, l=l, seed=seed, conditional=conditional, schedule=sched,
                              frame_stride=frame_stride))
        if last is not None:
            arrs["last"] = last
    else:
        arrs.update(ret_img=ret, last=last,
                    meta=meta(cfg, B=B, r=r, l=l, seed=seed, conditional=conditional, schedule=sched))
    save(name, **arrs)
    fin = ret[-B:]
    print(f"    ({time.time() - t0:.1f}s, final std {fin.std():.3f}, saturated {np.mean(np.abs(fin) >= 1):.2%})")


def only(name):
    """python tests/golden/make_golden.py [fixture.npz ...] regenerates just the named fixtures."""
    return len(sys.argv) < 2 or name in sys.argv[1:]


if __name__ == "__main__":
    tiny = synth.tiny_unet_config()
    s100 = {"schedule": "linear", "n_timestep": 100, "linear_start": 1e-6, "linear_end": 1e-2}
    if only("schedules.npz"):
        print("schedules"); gen_schedules()
    if only("unet_tiny.npz"):
        print("unet tiny"); gen_unet("unet_tiny.npz", tiny, B=2, r=16, seed=1, with_taps=True)
    if only("unet_yml224_r16.npz"):
        print("unet yml-literal r=16"); gen_unet("unet_yml224_r16.npz", synth.yml_unet_config(224), B=2, r=16, seed=2)
    if only("unet_yml128_r32.npz"):
        print("unet 128-variant r=32"); gen_unet("unet_yml128_r32.npz", synth.yml_unet_config(128), B=1, r=32, seed=3)
    if only("unet_yml224_r128.npz"):
        print("unet yml-literal r=128"); gen_unet("unet_yml224_r128.npz", synth.yml_unet_config(224), B=1, r=128, seed=4)
    if only("unet_yml128_r128.npz"):
        # BASELINE.json config 3 ("attention-heavy"): image_size=128 puts attention at the 16x16 level
        # (N = 256 tokens x 5 modules + mid), reference placement logic unet.py:192-207
        print("unet 128-variant r=128"); gen_unet("unet_yml128_r128.npz", synth.yml_unet_config(128), B=1, r=128, seed=8)
    if only("sampler_tiny.npz"):
        s20 = {"schedule": "linear", "n_timestep": 20, "linear_start": 1e-4, "linear_end": 2e-2}
        print("sampler tiny"); gen_sampler("sampler_tiny.npz", tiny, s20, B=2, r=16, l=8, seed=5)
    if only("sampler_uncond_tiny.npz"):
        tiny_u = graph.UNetConfig(in_channel=3, out_channel=3, inner_channel=32, channel_mults=(1, 2),
                                  attn_res=(8,), res_blocks=1, dropout=0.0, image_size=16)
        s10 = {"schedule": "cosine", "n_timestep": 10, "linear_start": 1e-4, "linear_end": 2e-2}
        print("sampler tiny unconditional"); gen_sampler("sampler_uncond_tiny.npz", tiny_u, s10, B=2, r=16, l=0, seed=6, conditional=False)
    if only("sampler_cfg1_8_16.npz"):
        # BASELINE.json config 1: sr_sr3_VGGF2_8_16, batch 4, 100-step DDPM (config/sr_sr3_VGGF2_8_16_model2.yml:52-57)
        print("sampler config 1"); gen_sampler("sampler_cfg1_8_16.npz", synth.yml_unet_config(224), s100, B=4, r=16, l=8, seed=7)
    if only("sampler_cfg5_32_128.npz"):
        # the benchmarked resolution over a whole schedule: BASELINE.json config 5's SR stage
        # (sr_sr3_VGGF2_32_128: 32 -> 128, T = 100, config/sr_sr3_VGGF2_32_128_model3.yml), B = 2,
        # yml-literal UNet; intermediate frames stored sub-sampled (every 4th pixel per axis)
        print("sampler config 5 (128x128, T=100)")
        gen_sampler("sampler_cfg5_32_128.npz", synth.yml_unet_config(224), s100, B=2, r=128, l=32, seed=9, frame_stride=4)
    if only("sampler_cfg2_16_128_T1000.npz"):
        # BASELINE.json config 2, the benchmarked workload, over its FULL horizon: 16 -> 128, yml-literal UNet,
        # T = 1000 linear 1e-6..1e-2 (config/sr_sr3_VGGF2_8_128_model3.yml's n_timestep: 1000), B = 1;
        # diffusion.py:189-215 run once (continous=True); intermediate frames sub-sampled (every 4th pixel)
        s1000 = {"schedule": "linear", "n_timestep": 1000, "linear_start": 1e-6, "linear_end": 1e-2}
        print("sampler config 2 (16 -> 128, T=1000)")
        gen_sampler("sampler_cfg2_16_128_T1000.npz", synth.yml_unet_config(224), s1000, B=1, r=128, l=16, seed=10,
                    frame_stride=4, with_last=False)
