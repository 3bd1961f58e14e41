"""Host logic that needs no GPU: parameter inventory, schedule, C-ABI symbol table, config
mirroring, metrics. (Checked against the reference-made golden vectors.)"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import REPO, cfg_from_meta, load_golden, pkg

graph = pkg("graph")
schedule = pkg("schedule")
synth = pkg("synth")
metrics = pkg("metrics")


@pytest.mark.parametrize("name", ["unet_tiny.npz", "unet_yml224_r16.npz", "unet_yml128_r32.npz"])
def test_param_specs_equal_reference_state_dict(name):
    g = load_golden(name)
    cfg = cfg_from_meta(g["meta"])
    got = [(n, list(s)) for n, s, _ in graph.param_specs(cfg)]
    assert got == [(k, s) for k, s in g["state_dict_keys"]]
    assert graph.count_params(cfg) == g["meta"]["n_params"]


def test_survey_numbers():
    c224, c128 = synth.yml_unet_config(224), synth.yml_unet_config(128)
    assert graph.count_params(c224) == 92_556_931
    assert graph.count_params(c128) == 97_807_491
    assert abs(graph.flops_per_image(c224, 128, 128) / 1e9 - 89.00) < 0.01
    assert abs(graph.flops_per_image(c128, 128, 128) / 1e9 - 92.35) < 0.01
    assert abs(graph.flops_per_image(c224, 16, 16) / 1e9 - 1.392) < 0.001


def test_schedule_bit_exact_vs_reference():
    g = load_golden("schedules.npz")
    for i, (s, T, a, b) in enumerate(g["cases"]):
        with np.errstate(divide="ignore", invalid="ignore"):
            bufs = schedule.schedule_buffers({"schedule": s, "n_timestep": T, "linear_start": a, "linear_end": b})
        for k in schedule.BUFFER_NAMES:
            np.testing.assert_array_equal(bufs[k], g[f"c{i}.{k}"], err_msg=f"{s} {k}")
        np.testing.assert_array_equal(bufs["sqrt_alphas_cumprod_prev"], g[f"c{i}.sqrt_alphas_cumprod_prev"])
        assert bufs["noise_level"].dtype == np.float32 and bufs["noise_level"].shape == (T + 1,)
    with pytest.raises(NotImplementedError):
        schedule.make_beta_schedule("bogus", 10)


def test_frame_steps_match_reference_bookkeeping():
    for T, n in [(100, 10), (1000, 10), (600, 10), (20, 7), (10, 10), (2000, 10), (7, 7)]:
        steps = schedule.frame_steps(T)
        assert len(steps) == n and steps[-1] == 0 and steps == sorted(steps, reverse=True)
    g = load_golden("sampler_cfg1_8_16.npz")
    assert g["ret_img"].shape[0] == g["meta"]["B"] * (1 + len(schedule.frame_steps(100)))


def test_library_exports_every_declared_symbol():
    """The C-ABI library loads on a GPU-less host and exports exactly what include/sr3hip.h
    declares (no compute call is made)."""
    lib_mod = pkg("_lib")
    if not os.path.exists(lib_mod.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = lib_mod.load()
    header = open(os.path.join(REPO, "include", "sr3hip.h")).read()
    declared = set(re.findall(r"\b(sr3_[a-z0-9_]+)\s*\(", header))
    declared -= {"sr3_ctx", "sr3_unet_cfg"}
    assert declared == set(lib_mod.PROTOTYPES), declared ^ set(lib_mod.PROTOTYPES)
    for name in declared:
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr)
    assert ctypes.sizeof(lib_mod.UnetCfg) == 100   # 25 x 4-byte fields of sr3_unet_cfg


def test_missing_library_fails_loudly(monkeypatch):
    lib_mod = pkg("_lib")
    monkeypatch.setattr(lib_mod, "_lib", None)
    monkeypatch.setattr(lib_mod, "LIB_PATH", "/nonexistent/libsr3hip.so")
    with pytest.raises(lib_mod.Sr3Error, match="no CPU fallback"):
        lib_mod.load()


def test_facade_is_storage_only_on_cpu():
    """define_G builds the reference's module tree and state_dict on a CPU-only host, loads a
    reference-shaped state_dict, and refuses to compute without a GPU."""
    import torch
    opt = synth.yml_opt(8, 16, 100)
    opt["sr"]["model"]["unet"].update(inner_channel=32, channel_multiplier=[1, 2], res_blocks=1, attn_res=[8])
    opt["sr"]["model"]["diffusion"]["image_size"] = 16
    netG = pkg().define_G(opt)
    cfg = netG.denoise_fn.cfg
    assert [k for k in netG.state_dict()] == ["denoise_fn." + n for n, _, _ in graph.param_specs(cfg)]
    sd = {"denoise_fn." + k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, 3).items()}
    res = netG.load_state_dict(sd, strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    netG.set_new_noise_schedule(opt["sr"]["model"]["beta_schedule"]["val"], ["cpu"])
    assert netG.num_timesteps == 100 and len(netG.state_dict()) == len(sd) + 12
    assert netG.train() is netG and netG.eval() is netG and len(list(netG.parameters())) == len(sd)
    netG.set_loss("cpu")
    with pytest.raises(pkg("_lib").Sr3Error):
        netG.super_resolution(torch.zeros(1, 3, 16, 16))
    with pytest.raises(NotImplementedError):
        netG({"HR": None, "SR": None})
    bad = dict(opt); bad["sr"] = {"model": dict(opt["sr"]["model"], which_model_G="ddpm")}
    with pytest.raises(NotImplementedError):
        pkg().define_G(bad)
    # phase == 'train' applies the orthogonal init of networks.py:47-58
    opt_t = dict(opt, phase="train")
    w = pkg().define_G(opt_t).denoise_fn.downs[1].res_block.block1.block[3].weight.detach().reshape(32, -1)
    np.testing.assert_allclose((w @ w.T).numpy(), np.eye(32), atol=1e-4)


def test_metrics_psnr_and_uint8():
    a = np.linspace(-1.2, 1.2, 3 * 4 * 4, dtype=np.float32).reshape(3, 4, 4)
    img = metrics.tensor2img(a)
    assert img.dtype == np.uint8 and img.shape == (4, 4, 3) and img.min() == 0 and img.max() == 255
    assert metrics.psnr(img, img) == float("inf")
    b = img.copy(); b[0, 0, 0] ^= 16
    assert abs(metrics.psnr(img, b) - 20 * np.log10(255 / np.sqrt(256 / img.size))) < 1e-9


def test_synth_is_deterministic():
    cfg = synth.tiny_unet_config()
    a, b = synth.synth_state_dict(cfg, 5), synth.synth_state_dict(cfg, 5)
    assert all(np.array_equal(a[k], b[k]) for k in a) and not np.array_equal(
        a["downs.0.weight"], synth.synth_state_dict(cfg, 6)["downs.0.weight"])
    c = synth.synth_cond(2, 16, 8, 1)
    assert c.shape == (2, 3, 16, 16) and c.dtype == np.float32 and np.abs(c).max() <= 1
    assert synth.synth_noise(3, 2, 3, 4, 4, 9).shape == (3, 2, 3, 4, 4)


def test_philox_twin_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors): zero and all-ones."""
    import philox
    z = philox.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in z] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = 0xFFFFFFFF
    o = philox.philox4x32_10(f, f, f, f, f, f)
    assert [int(x) for x in o] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]


def test_ssim_matches_independent_scipy_evaluation():
    """validation.ssim (numpy restatement of core/metrics.py:84-104) against scipy's correlate."""
    from scipy.ndimage import correlate
    val = pkg("validation")
    rs = np.random.RandomState(0)
    a = rs.randint(0, 256, (40, 48)).astype(np.uint8)
    b = np.clip(a.astype(int) + rs.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
    k = val.gaussian_kernel(11, 1.5)
    assert abs(k.sum() - 1) < 1e-12 and k.argmax() == 5
    win = np.outer(k, k)

    def f(x):
        return correlate(x.astype(np.float64), win, mode="mirror")[5:-5, 5:-5]   # filter2D default border, cropped
    C1, C2 = 6.5025, 58.5225
    A, B = a.astype(np.float64), b.astype(np.float64)
    mu1, mu2 = f(A), f(B)
    ref = (((2 * mu1 * mu2 + C1) * (2 * (f(A * B) - mu1 * mu2) + C2)) /
           ((mu1 ** 2 + mu2 ** 2 + C1) * (f(A * A) - mu1 ** 2 + f(B * B) - mu2 ** 2 + C2))).mean()
    assert abs(val.ssim(a, b) - ref) < 1e-10
    assert abs(val.ssim(a, a) - 1.0) < 1e-12
    rgb_a, rgb_b = np.stack([a, b, a], -1), np.stack([b, b, a], -1)
    assert abs(val.calculate_ssim(rgb_a, rgb_b) - (val.ssim(a, b) + 2) / 3) < 1e-12
    with pytest.raises(ValueError):
        val.calculate_ssim(a, b[:-1])


def test_chunk_plan_equal_chunks():
    """Batches above one library call's limit (sr3_max_batch: every activation tensor < 4 GiB) are split by the facade into
    the fewest EQUAL chunks, the last one padded (diffusion.GaussianDiffusion.chunk_plan): one workspace, one captured graph and
    the same kernels for every chunk (lib/trainer_temp.py:441-446 stacks 15 samples x N images; configs[3] is 512 images)."""
    G = pkg("diffusion").GaussianDiffusion
    assert G.chunk_plan(64, 330) == (1, 64)
    assert G.chunk_plan(330, 330) == (1, 330)
    assert G.chunk_plan(331, 330) == (2, 166)
    assert G.chunk_plan(400, 330) == (2, 200)
    assert G.chunk_plan(512, 330) == (2, 256)
    assert G.chunk_plan(960, 250) == (4, 240)
    for B in range(1, 1200, 7):
        for limit in (1, 3, 100, 330):
            n, chunk = G.chunk_plan(B, limit)
            assert chunk <= max(limit, 1) and n * chunk >= B and (n - 1) * chunk < B
            assert n == -(-B // limit)              # fewest chunks
