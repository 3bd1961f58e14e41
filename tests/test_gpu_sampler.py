"""p_sample_loop on the GPU (sr3_sample through the C-ABI) against golden runs of the reference's
own sampler with identical weights, schedule and injected noise. Bar: <= 1e-3 max-abs fp32
(BASELINE.json north_star), plus PSNR as core/metrics.py defines it."""
import numpy as np
import pytest

import sr3_oracle as oracle
from conftest import cfg_from_meta, load_golden, pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")
schedule = pkg("schedule")
metrics = pkg("metrics")
BAR = 1e-3


PRECISIONS = ["f32", "f16x3"]


def _engine(cfg, seed, sched_opt, prec="f32"):
    e = pkg("engine").Engine(cfg, 0)
    e.load_state_dict(synth.synth_state_dict(cfg, seed))
    e.set_precision(prec)
    with np.errstate(divide="ignore", invalid="ignore"):
        e.set_schedule(schedule.schedule_buffers(sched_opt))
    return e


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("name", ["sampler_tiny.npz", "sampler_uncond_tiny.npz", "sampler_cfg1_8_16.npz"])
def test_sampler_golden(name, prec):
    g = load_golden(name)
    m = g["meta"]
    cfg = cfg_from_meta(m)
    B, r, T = m["B"], m["r"], m["schedule"]["n_timestep"]
    e = _engine(cfg, m["seed"], m["schedule"], prec)
    noise = synth.synth_noise(T, B, 3, r, r, m["seed"])
    cond = g["cond"] if m["conditional"] else None
    final, frames = e.sample_np(cond, noise=noise, frames=True, shape=(B, 3, r, r))
    nf = frames.shape[0]
    assert nf == len(schedule.frame_steps(T)) == (g["ret_img"].shape[0] // B - 1)
    ref_frames = g["ret_img"][B:].reshape(nf, B, 3, r, r)
    err = np.abs(frames - ref_frames).reshape(nf, -1).max(1)
    print(f"{name} [{prec}]: per-frame max abs err {np.array2string(err, precision=2)}; "
          f"PSNR final {metrics.batch_psnr(final, ref_frames[-1]):.1f} dB")
    assert err.max() <= BAR
    np.testing.assert_array_equal(final, frames[-1])
    assert np.abs(final[-1] - g["last"]).max() <= BAR     # the reference's non-continuous return
    e.close()


def test_sampler_philox_equals_injected_twin():
    """Device RNG path == injected-noise path fed with the CPU twin of the same stream, and the
    image_offset makes a shard reproduce its slice of the full batch."""
    import philox
    cfg = synth.tiny_unet_config()
    sched = {"schedule": "linear", "n_timestep": 12, "linear_start": 1e-4, "linear_end": 2e-2}
    e = _engine(cfg, 31, sched)
    B, r, T, seed = 3, 16, 12, 987654321
    cond = synth.synth_cond(B, r, 8, 31)
    a = e.sample_np(cond, seed=seed)
    b = e.sample_np(cond, noise=philox.noise_slabs(seed, T, B, 3, r, r))
    assert np.abs(a - b).max() < 2e-4
    shard = e.sample_np(cond[1:], seed=seed, image_offset=1)
    np.testing.assert_allclose(shard, a[1:], atol=1e-6)
    assert np.abs(e.sample_np(cond, seed=seed + 1) - a).max() > 1e-2
    e.close()


def test_step_api_equals_sample():
    cfg = synth.tiny_unet_config()
    sched = {"schedule": "linear", "n_timestep": 6, "linear_start": 1e-4, "linear_end": 2e-2}
    e = _engine(cfg, 41, sched)
    B, r, T = 2, 16, 6
    cond, noise = synth.synth_cond(B, r, 8, 41), synth.synth_noise(T, B, 3, r, r, 41)
    want = e.sample_np(cond, noise=noise)
    dc, dn, out = e.to_device(cond), e.to_device(noise), e.buffer(B * 3 * r * r)
    slab = B * 3 * r * r * 4
    e.sample_begin(dc.ptr, B, r, r, dn.ptr)
    for t in reversed(range(T)):
        e.sample_step(t, dn.ptr + (T - t) * slab if t > 0 else None)
    e.sample_end(out.ptr)
    np.testing.assert_array_equal(out.download((B, 3, r, r)), want)
    with pytest.raises(pkg("_lib").Sr3Error, match="outside schedule"):
        e.sample_step(T)
    e.close()


def test_torch_facade_super_resolution():
    """GaussianDiffusion.super_resolution(x_in, continous) return conventions (diffusion.py:212-215)."""
    import torch
    g = load_golden("sampler_tiny.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    opt = {"phase": "val", "sr": {"model": {
        "which_model_G": "sr3",
        "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": cfg.inner_channel,
                 "channel_multiplier": list(cfg.channel_mults), "attn_res": list(cfg.attn_res),
                 "res_blocks": cfg.res_blocks, "dropout": 0.0},
        "beta_schedule": {"train": m["schedule"], "val": m["schedule"]},
        "diffusion": {"image_size": cfg.image_size, "channels": 3, "conditional": True}}}}
    netG = pkg().define_G(opt).cuda()
    netG.load_state_dict({"denoise_fn." + k: torch.from_numpy(v)
                          for k, v in synth.synth_state_dict(cfg, m["seed"]).items()}, strict=False)
    netG.set_new_noise_schedule(m["schedule"], [0])
    assert len([k for k in netG.state_dict() if not k.startswith("denoise_fn.")]) == 12
    B, r, T = m["B"], m["r"], m["schedule"]["n_timestep"]
    noise = torch.from_numpy(synth.synth_noise(T, B, 3, r, r, m["seed"]))
    cond = torch.from_numpy(g["cond"]).cuda()
    ret = netG.p_sample_loop(cond, continous=True, noise=noise)
    assert tuple(ret.shape) == g["ret_img"].shape
    assert np.abs(ret.cpu().numpy() - g["ret_img"]).max() <= BAR
    last = netG.p_sample_loop(cond, continous=False, noise=noise)
    assert tuple(last.shape) == (3, r, r)
    assert np.abs(last.cpu().numpy() - g["last"]).max() <= BAR
    # seeded device RNG: reproducible under torch.manual_seed, different across seeds
    torch.manual_seed(3); a = netG.super_resolution_batch(cond)
    torch.manual_seed(3); b = netG.super_resolution_batch(cond)
    torch.manual_seed(4); c = netG.super_resolution_batch(cond)
    assert torch.equal(a, b) and not torch.equal(a, c) and tuple(a.shape) == (B, 3, r, r)
    with pytest.raises(NotImplementedError):
        netG({"HR": cond, "SR": cond})
