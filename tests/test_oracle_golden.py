"""Pins the CPU oracle (oracle/sr3_oracle.py) to vectors produced by the reference itself
(tests/golden/make_golden.py). CPU only."""
import numpy as np
import pytest

import sr3_oracle as oracle
import sr3_oracle_aten as aten
from conftest import cfg_from_meta, load_golden, pkg

synth = pkg("synth")

BUFS = ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
        "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
        "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
        "posterior_mean_coef1", "posterior_mean_coef2")


def test_schedule_bit_exact():
    g = load_golden("schedules.npz")
    for i, (s, T, a, b) in enumerate(g["cases"]):
        with np.errstate(divide="ignore", invalid="ignore"):
            sch = oracle.noise_schedule({"schedule": s, "n_timestep": T, "linear_start": a, "linear_end": b})
        for k in BUFS:
            np.testing.assert_array_equal(sch[k], g[f"c{i}.{k}"], err_msg=f"{s} T={T} {k}")
        np.testing.assert_array_equal(sch["sqrt_alphas_cumprod_prev"], g[f"c{i}.sqrt_alphas_cumprod_prev"])


def test_unknown_schedule_raises():
    with pytest.raises(NotImplementedError):
        oracle.make_beta_schedule("nope", 10)


@pytest.mark.parametrize("name,tol", [("unet_tiny.npz", 2e-5), ("unet_yml224_r16.npz", 2e-5),
                                      ("unet_yml128_r32.npz", 2e-5), ("unet_yml224_r128.npz", 2e-5),
                                      ("unet_yml128_r128.npz", 2e-5)])
def test_unet_forward_matches_reference(name, tol):
    g = load_golden(name)
    cfg = cfg_from_meta(g["meta"])
    sd = synth.synth_state_dict(cfg, g["meta"]["seed"])
    taps = {}
    eps = oracle.unet_forward(sd, cfg, g["x"], g["noise_level"], taps=taps)
    for k in g:
        if k.startswith("tap."):
            np.testing.assert_allclose(taps[k[4:]], g[k], atol=tol, rtol=0, err_msg=k)
    np.testing.assert_allclose(eps, g["eps"], atol=tol, rtol=0)


@pytest.mark.parametrize("name", ["sampler_tiny.npz", "sampler_uncond_tiny.npz", "sampler_cfg1_8_16.npz"])
def test_sampler_matches_reference(name):
    g = load_golden(name)
    m = g["meta"]
    cfg = cfg_from_meta(m)
    sd = synth.synth_state_dict(cfg, m["seed"])
    sch = oracle.noise_schedule(m["schedule"])
    B, r, T = m["B"], m["r"], m["schedule"]["n_timestep"]
    noise = synth.synth_noise(T, B, 3, r, r, m["seed"])
    cond = g["cond"] if m["conditional"] else None
    if cond is not None:
        np.testing.assert_array_equal(cond, synth.synth_cond(B, r, m["l"], m["seed"]))
    final, frames = oracle.p_sample_loop(sd, cfg, sch, cond, noise)
    first = cond if cond is not None else noise[0]
    ret = np.concatenate([first, frames.reshape(-1, *frames.shape[2:])], axis=0)
    assert ret.shape == g["ret_img"].shape
    np.testing.assert_allclose(ret, g["ret_img"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(final[-1], g["last"], atol=1e-4, rtol=0)


def test_sampler_128px_head_matches_reference():
    """BASELINE config 5's SR stage at the benchmarked resolution (32 -> 128, T = 100, B = 2): the
    oracle is run up to the SECOND recorded frame (frames follow i % (1 | T // 10) == 0, diffusion.py:192,
    i.e. after t = 99 and t = 88: 12 of the 100 steps, ~40 s of numpy at 128x128; the whole loop is the
    GPU test's job) and compared with the reference's frames."""
    g = load_golden("sampler_cfg5_32_128.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    sd = synth.synth_state_dict(cfg, m["seed"])
    sch = oracle.noise_schedule(m["schedule"])
    B, r, T, st = m["B"], m["r"], m["schedule"]["n_timestep"], m["frame_stride"]
    noise = synth.synth_noise(T, B, 3, r, r, m["seed"])
    np.testing.assert_array_equal(g["cond"], synth.synth_cond(B, r, m["l"], m["seed"]))
    assert g["frames_sub"].shape == (10, B, 3, r // st, r // st) and g["final"].shape == (B, 3, r, r)
    np.testing.assert_array_equal(g["final"][..., ::st, ::st], g["frames_sub"][-1])
    np.testing.assert_array_equal(g["final"][-1], g["last"])
    si = 1 | (T // 10)
    x, f = noise[0], 0
    for k, t in enumerate(range(T - 1, T - 13, -1)):
        x = oracle.p_sample(sd, cfg, sch, x, t, g["cond"], noise[k + 1])
        if t % si == 0:
            np.testing.assert_allclose(x[..., ::st, ::st], g["frames_sub"][f], atol=1e-4, rtol=0, err_msg=f"frame {f} (t={t})")
            f += 1
    assert f == 2


@pytest.mark.parametrize("name,tol", [("unet_tiny.npz", 2e-5), ("unet_yml224_r16.npz", 2e-5),
                                      ("unet_yml128_r32.npz", 2e-5), ("unet_yml224_r128.npz", 2e-5),
                                      ("unet_yml128_r128.npz", 2e-5)])
def test_aten_unet_forward_matches_reference(name, tol):
    """oracle/sr3_oracle_aten.py (torch CPU operators; bench.py's `cpu_baseline` leg) against the same
    reference-made fixtures as the numpy oracle."""
    import torch
    g = load_golden(name)
    cfg = cfg_from_meta(g["meta"])
    sd = aten.to_torch_state(synth.synth_state_dict(cfg, g["meta"]["seed"]))
    taps = {}
    with torch.no_grad():
        eps = aten.unet_forward(sd, cfg, torch.from_numpy(g["x"]), torch.from_numpy(g["noise_level"]), taps=taps).numpy()
    for k in g:
        if k.startswith("tap."):
            np.testing.assert_allclose(taps[k[4:]], g[k], atol=tol, rtol=0, err_msg=k)
    np.testing.assert_allclose(eps, g["eps"], atol=tol, rtol=0)


@pytest.mark.parametrize("name", ["sampler_tiny.npz", "sampler_uncond_tiny.npz", "sampler_cfg1_8_16.npz"])
def test_aten_sampler_matches_reference(name):
    g = load_golden(name)
    m = g["meta"]
    cfg = cfg_from_meta(m)
    sd = synth.synth_state_dict(cfg, m["seed"])
    sch = aten.noise_schedule(m["schedule"])
    B, r, T = m["B"], m["r"], m["schedule"]["n_timestep"]
    noise = synth.synth_noise(T, B, 3, r, r, m["seed"])
    cond = g["cond"] if m["conditional"] else None
    final, frames = aten.p_sample_loop(sd, cfg, sch, cond, noise)
    first = cond if cond is not None else noise[0]
    ret = np.concatenate([first, frames.reshape(-1, *frames.shape[2:])], axis=0)
    np.testing.assert_allclose(ret, g["ret_img"], atol=1e-4, rtol=0)
    np.testing.assert_allclose(final[-1], g["last"], atol=1e-4, rtol=0)


def test_sampler_cfg2_T1000_head_matches_reference():
    """BASELINE config 2 (the benchmarked workload: 16 -> 128, yml-literal UNet, T = 1000, B = 1) pinned to
    the reference over its full horizon by tests/golden/sampler_cfg2_16_128_T1000.npz; the whole loop is the
    GPU tests' job (tests/test_gpu_round3.py). Here both CPU oracles reproduce its head: the aten oracle runs
    to the first recorded frame (t = 999 ... 909 — frames follow i % (1 | T // 10) == 0 with si = 101,
    diffusion.py:192,210 — 91 steps), the numpy oracle the first 6 steps against the aten state."""
    import torch
    g = load_golden("sampler_cfg2_16_128_T1000.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    sd = synth.synth_state_dict(cfg, m["seed"])
    sch = oracle.noise_schedule(m["schedule"])
    B, r, T, st = m["B"], m["r"], m["schedule"]["n_timestep"], m["frame_stride"]
    assert (B, r, T, m["l"]) == (1, 128, 1000, 16)
    noise = synth.synth_noise(T, B, 3, r, r, m["seed"])
    np.testing.assert_array_equal(g["cond"], synth.synth_cond(B, r, m["l"], m["seed"]))
    assert g["frames_sub"].shape == (10, B, 3, r // st, r // st) and g["final"].shape == (B, 3, r, r)
    np.testing.assert_array_equal(g["final"][..., ::st, ::st], g["frames_sub"][-1])
    si = 1 | (T // 10)
    tsd = aten.to_torch_state(sd)
    tc = torch.from_numpy(g["cond"])
    x = torch.from_numpy(noise[0].copy())
    xn = noise[0]
    with torch.no_grad():
        for k, t in enumerate(range(T - 1, T - 1 - 91, -1)):
            x = aten.p_sample(tsd, cfg, sch, x, t, tc, torch.from_numpy(noise[k + 1].copy()))
            if k < 6:
                xn = oracle.p_sample(sd, cfg, sch, xn, t, g["cond"], noise[k + 1])
                np.testing.assert_allclose(xn, x.numpy(), atol=2e-5, rtol=0, err_msg=f"numpy vs aten oracle at t={t}")
    assert t % si == 0
    np.testing.assert_allclose(x.numpy()[..., ::st, ::st], g["frames_sub"][0], atol=1e-4, rtol=0)


def test_pil_bicubic_restatement_bit_exact():
    """oracle/pil_bicubic.py against Pillow's own output (tests/golden/make_golden_preproc.py)."""
    import pil_bicubic as pb
    g = load_golden("preproc_bicubic.npz")
    for i, (a, b) in enumerate(g["meta"]["cases"]):
        np.testing.assert_array_equal(pb.resize_u8(g[f"in{i}"], b, b), g[f"out{i}"], err_msg=f"{a}->{b}")
    lr = pb.resize_u8(g["chain_hr"], 16, 16)
    np.testing.assert_array_equal(lr, g["chain_lr"])
    np.testing.assert_array_equal(pb.resize_u8(lr, 128, 128), g["chain_sr"])
    t = pb.to_tensor_pm1(g["chain_sr"])
    assert t.shape == (3, 128, 128) and t.dtype == np.float32 and t.min() >= -1 and t.max() <= 1


def test_post_oracle_self_consistency():
    """oracle/post_oracle.py. No cv2 fixture exists (parity unpinned), so the
    restated fixed-point resize is held to what it approximates: within 1 grey level of the exact
    bilinear value, exact on constant images, and identity-like at scale 1; tensor2img is checked
    on hand-computed values (round half to even, clamp)."""
    import post_oracle as post
    rs = np.random.RandomState(0)
    img = rs.randint(0, 256, (128, 128, 3)).astype(np.uint8)
    for size in (224, 200, 131):
        up = post.cv2_resize_linear_u8(img, size, size)
        assert up.shape == (size, size, 3) and up.dtype == np.uint8
        assert np.abs(up.astype(np.float64) - post.float_bilinear_u8(img, size, size)).max() <= 1.0
    flat = np.full((16, 16, 3), 201, np.uint8)
    assert (post.cv2_resize_linear_u8(flat, 224, 224) == 201).all()
    x = np.zeros((3, 2, 2), np.float32)
    x[0, 0, 0], x[0, 0, 1], x[0, 1, 0], x[0, 1, 1] = -3.0, 3.0, 1.0 / 255, 3.0 / 255   # 0, 255, 128 (127.5+.5 -> even 128), 129 (128.5+.5)
    got = post.tensor2img(x)[:, :, 0]
    assert got.tolist() == [[0, 255], [128, 129]]
    blob = post.cv2_blob_from_image(post.cv2_resize_linear_u8(img, 224, 224), 112)
    assert blob.shape == (3, 112, 112) and blob.dtype == np.float32
    area = post.cv2_resize_linear_u8(img, 224, 224).astype(np.int64)
    r00 = (area[0, 0, 0] + area[0, 1, 0] + area[1, 0, 0] + area[1, 1, 0] + 2) >> 2
    assert blob[2, 0, 0] == np.float32((np.float32(r00) - np.float32(127.5)) * np.float32(1 / 127.5))   # swapRB: R is channel 2
    tb = post.tensor_blob_torch(rs.uniform(-1, 1, (1, 3, 128, 128)).astype(np.float32))
    assert tb.shape == (1, 3, 112, 112) and np.abs(tb).max() <= 1.0 + 1e-6
