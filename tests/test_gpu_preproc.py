"""Pre-processing kernel (sr3_preprocess_bicubic) against Pillow's own output (golden) and the
oracle restatement: bit-exact uint8, exact fp32 tensor; plus the validation driver on the GPU."""
import numpy as np
import pytest

import pil_bicubic as pb
from conftest import load_golden, pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")


@pytest.fixture(scope="module")
def eng():
    e = pkg("engine").Engine(synth.tiny_unet_config(), 0)
    yield e
    e.close()


def test_bicubic_bit_exact_vs_pillow(eng):
    g = load_golden("preproc_bicubic.npz")
    for i, (a, b) in enumerate(g["meta"]["cases"]):
        t, u8 = eng.preprocess_bicubic_np(g[f"in{i}"][None], b, b)
        np.testing.assert_array_equal(u8[0], g[f"out{i}"], err_msg=f"{a}->{b}")
        np.testing.assert_array_equal(t[0], pb.to_tensor_pm1(g[f"out{i}"]))
    # the reference chain HR 128 -> LR 16 -> SR 128 (prepare_data.py:37-47)
    _, lr = eng.preprocess_bicubic_np(g["chain_hr"][None], 16, 16)
    np.testing.assert_array_equal(lr[0], g["chain_lr"])
    t, sr = eng.preprocess_bicubic_np(lr, 128, 128)
    np.testing.assert_array_equal(sr[0], g["chain_sr"])
    assert t.shape == (1, 3, 128, 128) and t.dtype == np.float32


def test_bicubic_batch_ragged_and_identity(eng):
    rs = np.random.RandomState(4)
    img = rs.randint(0, 256, (5, 24, 40, 3)).astype(np.uint8)        # non-square, batch 5
    t, u8 = eng.preprocess_bicubic_np(img, 57, 33)
    for b in range(5):
        np.testing.assert_array_equal(u8[b], pb.resize_u8(img[b], 57, 33))
    t, u8 = eng.preprocess_bicubic_np(img, 24, 40)                   # same size: pure conversion
    np.testing.assert_array_equal(u8, img)
    np.testing.assert_array_equal(t[2], pb.to_tensor_pm1(img[2]))
    t, u8 = eng.preprocess_bicubic_np(img, 24, 80)                   # horizontal pass only
    np.testing.assert_array_equal(u8[1], pb.resize_u8(img[1], 24, 80))


def test_validation_driver(eng):
    """images x samples as one batch through the sampler, PSNR/SSIM per (sample, image)."""
    import torch
    val = pkg("validation")
    cfg = synth.tiny_unet_config()
    sched = {"schedule": "linear", "n_timestep": 8, "linear_start": 1e-4, "linear_end": 2e-2}
    opt = {"phase": "val", "sr": {"model": {
        "which_model_G": "sr3",
        "unet": {"in_channel": 6, "out_channel": 3, "inner_channel": cfg.inner_channel,
                 "channel_multiplier": list(cfg.channel_mults), "attn_res": list(cfg.attn_res),
                 "res_blocks": cfg.res_blocks, "dropout": 0.0},
        "beta_schedule": {"train": sched, "val": sched},
        "diffusion": {"image_size": cfg.image_size, "channels": 3, "conditional": True}}}}
    netG = pkg().define_G(opt).cuda()
    netG.load_state_dict({"denoise_fn." + k: torch.from_numpy(v)
                          for k, v in synth.synth_state_dict(cfg, 8).items()}, strict=False)
    netG.set_new_noise_schedule(sched, [0])
    N, K = 3, 2
    sr = torch.from_numpy(synth.synth_cond(N, 16, 8, 8)).cuda()
    r = val.validate_batch(netG, sr, sr, samples=K, seed=5)
    assert r["psnr"].shape == (K, N) and r["ssim"].shape == (K, N) and tuple(r["images"].shape) == (K * N, 3, 16, 16)
    assert np.isfinite(r["mean_psnr"]) and -1 <= r["mean_ssim"] <= 1
    # sample k of image i is row k*N + i, and equals running image i alone with that global index
    alone = netG.super_resolution_batch(sr[1:2], seed=5, image_offset=1 * N + 1)
    assert torch.allclose(alone[0], r["images"][1 * N + 1], atol=1e-5)
    # scoring the sampler's own output against itself: PSNR inf, SSIM 1
    r2 = val.validate_batch(netG, sr, r["images"][:N], samples=1, seed=5)
    assert np.isinf(r2["psnr"]).all() and np.allclose(r2["ssim"], 1.0)
