"""The 'f16f8' arithmetic (sr3_set_precision(ctx, 2); include/sr3hip.h): split-f16 with the two correction products
x_lo*w_hi + x_hi*w_lo of the MFMA-bound 3x3 convs on the fp8 matrix path (ConvParams::f8, kernels_conv.hip). Those
convs are the full-batch 32x32- and 16x16-pixel levels, so every case here runs at B = 32..64; reference-made
fixtures (B = 1 or 2) are replicated along the batch and every replica is held to the reference's output.

Bars: single conv 2e-4 absolute on O(1) outputs (observed 5.5e-5..6.1e-5; the CPU emulation of the same operand
formats, tests/emulate_operand_formats.py, predicts 5.1e-5); UNet forward 5e-4 (observed 1e-5); samplers the north-star
1e-3 (observed 9e-6 over T = 100, 1.9e-5 over the headline T = 1000 run).
The reference computes in plain fp32 (model/sr/sr3_modules/unet.py:235-265, diffusion.py:189-215)."""
import numpy as np
import pytest

import sr3_oracle as oracle
from conftest import cfg_from_meta, load_golden, pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")
schedule = pkg("schedule")
Sr3Error = pkg("_lib").Sr3Error
BAR = 1e-3


@pytest.fixture(scope="module")
def eng():
    e = pkg("engine").Engine(synth.tiny_unet_config(), 0)
    e.load_state_dict(synth.synth_state_dict(e.cfg, 11))
    yield e
    e.close()


def _rand(rs, *shape):
    return rs.standard_normal(shape).astype(np.float32)


def test_which_convs_take_the_fp8_path(eng):
    ok = eng.conv_f8_supported
    assert ok(64, 32, 32, 256, 256) and ok(64, 16, 16, 512, 512) and ok(64, 16, 16, 512, 1024) and ok(32, 32, 32, 256, 768)
    assert not ok(64, 64, 64, 128, 128)       # operand-movement bound levels keep f16x3
    assert not ok(64, 128, 128, 64, 64)
    assert not ok(64, 8, 8, 512, 512)         # the 8x8 level runs the in-place split-K kernel
    assert not ok(4, 32, 32, 256, 256)        # few tiles: other tile shapes
    assert not ok(64, 32, 32, 64, 64)         # 128x64 tile


F8_CASES = [
    # B, H, W, C0, C1, Cout
    (64, 32, 32, 128, 0, 256),
    (64, 16, 16, 256, 0, 512),
    (64, 16, 16, 128, 128, 512),      # concatenated input (x || skip through one apply pass)
    (32, 32, 32, 96, 0, 256),         # three K chunks
    (64, 16, 32, 64, 0, 512),         # non-square
]


@pytest.mark.parametrize("case", F8_CASES)
def test_conv2d_f16f8(eng, case):
    B, H, W, C0, C1, Cout = case
    assert eng.conv_f8_supported(B, H, W, Cout, C0 + C1)
    rs = np.random.RandomState(hash(case) & 0xFFFF)
    x0 = _rand(rs, B, H, W, C0)
    x1 = _rand(rs, B, H, W, C1) if C1 else None
    w = _rand(rs, Cout, C0 + C1, 3, 3) / np.sqrt((C0 + C1) * 9)
    b = _rand(rs, Cout)
    eng.set_precision("f16f8")
    got = eng.op_conv2d(x0, w, b, x1=x1)
    eng.set_precision("f16x3")
    ref3 = eng.op_conv2d(x0, w, b, x1=x1)
    eng.set_precision("f32")
    xin = x0 if x1 is None else np.concatenate([x0, x1], -1)
    want = oracle.conv2d(xin, w, b)
    e8, e3 = np.abs(got - want).max(), np.abs(ref3 - want).max()
    print(f"{case}: f16f8 {e8:.2e}  f16x3 {e3:.2e}")
    assert e3 < 2e-5 and e8 < 2e-4
    assert e8 > 2 * e3, "the fp8 path was not taken (error as small as f16x3's)"


def test_conv2d_f16f8_fused_prologue_epilogue(eng):
    """GroupNorm apply + Swish written straight in the F8C operand format, FeatureWiseAffine bias and residual in the
    epilogue (Block + noise_func + residual, unet.py:105-110)."""
    rs = np.random.RandomState(5)
    B, H, W, C, Cout = 64, 16, 16, 128, 512
    x = _rand(rs, B, H, W, C) * 2 + 0.5
    gamma, beta = 1 + 0.1 * _rand(rs, C), 0.1 * _rand(rs, C)
    w, b = _rand(rs, Cout, C, 3, 3) / np.sqrt(9 * C), _rand(rs, Cout)
    cb, resid = _rand(rs, B, Cout), _rand(rs, B, H, W, Cout)
    sc, sh = eng.op_groupnorm_affine(x, gamma, beta, 32)
    eng.set_precision("f16f8")
    got = eng.op_conv2d(x, w, b, gn_scale=sc, gn_shift=sh, swish=True, chan_bias=cb, resid=resid)
    eng.set_precision("f32")
    want = oracle.conv2d(oracle.swish(oracle.group_norm(x, gamma, beta, 32)), w, b) + cb[:, None, None, :] + resid
    err = np.abs(got - want).max()
    print(f"fused prologue/epilogue f16f8: {err:.2e}")
    assert err < 2e-4


def test_f16f8_range_limit_is_detected(eng):
    """e4m3 saturates beyond 448: such an activation raises the range flag (never a silent clamp)."""
    rs = np.random.RandomState(3)
    x = _rand(rs, 64, 16, 16, 64)
    x[5, 3, 7, 11] = 600.0
    w = _rand(rs, 512, 64, 3, 3) / 24
    eng.set_precision("f16f8")
    with pytest.raises(Sr3Error, match="range"):
        eng.op_conv2d(x, w, None)
    eng.set_precision("f16x3")
    got = eng.op_conv2d(x, w, None)             # f16x3 holds it (limit 65504)
    eng.set_precision("f32")
    assert np.abs(got - oracle.conv2d(x, w, None)).max() < 1e-4


def _engine(cfg, sd, prec, sched_opt=None):
    e = pkg("engine").Engine(cfg, 0)
    e.load_state_dict(sd)
    e.set_precision(prec)
    if sched_opt:
        e.set_schedule(schedule.schedule_buffers(sched_opt))
    return e


@pytest.mark.parametrize("name", ["unet_yml224_r128.npz", "unet_yml128_r128.npz"])
def test_unet_forward_golden_replicated_f16f8(name):
    """The reference-made UNet fixtures at 128x128 (yml-literal and the six-attention variant), the input replicated to
    B = 64 so that the 32x32 and 16x16 levels take the fp8 path; every replica against the reference's eps."""
    g = load_golden(name)
    cfg = cfg_from_meta(g["meta"])
    B = 64
    x = np.repeat(g["x"][:1], B, axis=0)
    nl = np.repeat(np.asarray(g["noise_level"]).reshape(-1)[:1], B)
    e = _engine(cfg, synth.synth_state_dict(cfg, g["meta"]["seed"]), "f16f8")
    eps8 = e.unet_forward_np(x, nl)
    e.set_precision("f16x3")
    eps3 = e.unet_forward_np(x, nl)
    e.close()
    e8 = np.abs(eps8 - g["eps"][:1]).max()
    e3 = np.abs(eps3 - g["eps"][:1]).max()
    print(f"{name} x{B}: f16f8 {e8:.2e}  f16x3 {e3:.2e} vs reference")
    assert e3 < 1e-4 and e8 < 5e-4
    assert e8 > 2 * e3, "the fp8 path was not taken"
    np.testing.assert_array_equal(eps8[0], eps8[B - 1])      # replicas are bit-identical


@pytest.mark.parametrize("B", [33, 63])
def test_unet_forward_ragged_batches_f16f8(B):
    """Shards that are not multiples of anything: at B = 33 / 63 only the 32x32 level has enough 128x128 tiles for the
    fp8 path (the 16x16 level stays f16x3); the forward equals the exact-f32 mode of the same library to 5e-4 and the
    images do not depend on their position in the batch."""
    cfg = synth.yml_unet_config(224)
    e = _engine(cfg, synth.synth_state_dict(cfg, 3), "f16f8")
    rs = np.random.RandomState(B)
    x = rs.standard_normal((B, 6, 128, 128)).astype(np.float32)
    x[B - 1] = x[0]
    nl = np.full(B, 0.37, np.float32)
    assert e.conv_f8_supported(B, 32, 32, 256, 256) and not e.conv_f8_supported(B, 16, 16, 512, 512)
    got = e.unet_forward_np(x, nl)
    e.set_precision("f32")
    want = e.unet_forward_np(x, nl)
    e.close()
    err = np.abs(got - want).max()
    print(f"B = {B}: f16f8 vs f32 {err:.2e}")
    assert err < 5e-4
    np.testing.assert_array_equal(got[0], got[B - 1])


def test_sampler_golden_128px_full_schedule_replicated_f16f8():
    """BASELINE config 5's SR stage over its whole schedule (32 -> 128, T = 100; tests/golden/sampler_cfg5_32_128.npz,
    a run of the reference itself with B = 2), replicated to B = 64 with the same injected noise per replica pair."""
    g = load_golden("sampler_cfg5_32_128.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    B0, r, T, st = m["B"], m["r"], m["schedule"]["n_timestep"], m["frame_stride"]
    rep = 64 // B0
    e = _engine(cfg, synth.synth_state_dict(cfg, m["seed"]), "f16f8", m["schedule"])
    noise = np.tile(synth.synth_noise(T, B0, 3, r, r, m["seed"]), (1, rep, 1, 1, 1))
    cond = np.tile(g["cond"], (rep, 1, 1, 1))
    final, frames = e.sample_np(cond, noise=noise, frames=True)
    again = e.sample_np(cond, noise=noise)
    np.testing.assert_array_equal(again, final)          # no atomics on data anywhere: repeats are bit-identical
    assert e.fallback_calls() == 0
    e.close()
    want_f = np.tile(g["frames_sub"], (1, rep, 1, 1, 1))
    err = np.abs(frames[..., ::st, ::st] - want_f).reshape(10, -1).max(1)
    e_fin = np.abs(final - np.tile(g["final"], (rep, 1, 1, 1))).max()
    print(f"cfg5 32->128 T=100 x{rep} [f16f8]: per-frame max abs err {np.array2string(err, precision=2)}; final {e_fin:.2e}")
    assert err.max() <= BAR and e_fin <= BAR
    np.testing.assert_array_equal(final[:B0], final[-B0:])


def test_fp8_range_falls_back_to_f16x3_first():
    """A network whose 32x32-level activations exceed the fp8 operand range (448) but not the fp16 range: the default
    policy finishes the call with all three products on the f16 path — bit-identical to the f16x3 mode — and warns; the
    strict policy fails. Forward and sampler (checkpointed replay inside sr3_sample)."""
    import warnings
    Sr3RangeWarning = pkg("_lib").Sr3RangeWarning
    cfg = synth.yml_unet_config(224)
    sd = synth.synth_state_dict(cfg, 3)
    name = next(k for k, v in sd.items() if k.startswith("downs.") and k.endswith("res_block.block2.block.0.weight") and v.shape == (256,))
    sd[name] = sd[name] * np.float32(400.0)          # GroupNorm gamma of a block2 at the 32x32 level: |activation| ~ 1e3
    B = 64
    rs = np.random.RandomState(1)
    x = rs.standard_normal((B, 6, 128, 128)).astype(np.float32)
    nl = np.full(B, 0.5, np.float32)
    e = _engine(cfg, sd, "f16x3", {"schedule": "linear", "n_timestep": 3, "linear_start": 1e-4, "linear_end": 2e-2})
    want = e.unet_forward_np(x, nl)
    cond = synth.synth_cond(B, 128, 16, 5)
    want_s = e.sample_np(cond, seed=7)
    assert e.fallback_calls() == 0
    e.set_precision("f16f8")
    with pytest.warns(Sr3RangeWarning, match="f16x3"):
        got = e.unet_forward_np(x, nl)
    np.testing.assert_array_equal(got, want)
    with pytest.warns(Sr3RangeWarning, match="f16x3"):
        got_s = e.sample_np(cond, seed=7)
    np.testing.assert_array_equal(got_s, want_s)
    assert e.fallback_calls() == 2
    e.set_range_policy(True)
    with pytest.raises(Sr3Error, match="range"):
        e.unet_forward_np(x, nl)
    e.set_range_policy(False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", Sr3RangeWarning)
        np.testing.assert_array_equal(e.unet_forward_np(x, nl), want)      # the mode is restored after a fallback
    e.close()


def test_headline_T1000_golden_replicated_f16f8():
    """The benchmarked configuration pinned to the reference over its FULL horizon with the fp8 path active: the
    reference-made fixture tests/golden/sampler_cfg2_16_128_T1000.npz (16 -> 128, yml-literal UNet, T = 1000, B = 1;
    diffusion.py:189-215) replicated to B = 64 — every replica takes the same injected noise (12.6 GB of it) — and each
    of the 64 results held to the reference's frames and final image. Bar 1e-3."""
    g = load_golden("sampler_cfg2_16_128_T1000.npz")
    m = g["meta"]
    cfg = cfg_from_meta(m)
    r, T, st = m["r"], m["schedule"]["n_timestep"], m["frame_stride"]
    assert (m["B"], r, T) == (1, 128, 1000)
    B = 64
    e = _engine(cfg, synth.synth_state_dict(cfg, m["seed"]), "f16f8", m["schedule"])
    noise = np.broadcast_to(synth.synth_noise(T, 1, 3, r, r, m["seed"]), (T, B, 3, r, r))
    cond = np.tile(g["cond"], (B, 1, 1, 1))
    final, frames = e.sample_np(cond, noise=np.ascontiguousarray(noise), frames=True)
    del noise
    assert e.fallback_calls() == 0
    e.close()
    err = np.abs(frames[..., ::st, ::st] - g["frames_sub"]).reshape(10, -1).max(1)       # (frames_sub broadcasts over B)
    e_fin = np.abs(final - g["final"]).max()
    print(f"cfg2 16->128 T=1000 x{B} [f16f8]: per-frame max abs err {np.array2string(err, precision=2)}; final {e_fin:.2e}")
    assert err.max() <= BAR and e_fin <= BAR
    np.testing.assert_array_equal(final[0], final[B - 1])


def test_headline_loop_f16f8_vs_f32():
    """The benchmarked configuration (16 -> 128, B = 64, T = 1000, device Philox noise): the whole loop in f16f8 against
    the exact-f32 mode of the same library — the same check tests/test_gpu_round2.py makes for f16x3 (the
    reference-made T = 1000 fixture is B = 1 and pins f16x3 / f32 in tests/test_gpu_round3.py)."""
    cfg = synth.yml_unet_config(224)
    sd = synth.synth_state_dict(cfg, 3)
    sched = {"schedule": "linear", "n_timestep": 1000, "linear_start": 1e-6, "linear_end": 1e-2}
    B = 64
    cond = synth.synth_cond(B, 128, 16, 5)
    e = _engine(cfg, sd, "f16f8", sched)
    out8 = e.sample_np(cond, seed=11)
    assert e.fallback_calls() == 0
    e.set_precision("f32")
    out32 = e.sample_np(cond[:8], seed=11)          # Philox is keyed by the global image index: images 0..7
    e.close()
    err = np.abs(out8[:8] - out32).max()
    print(f"headline loop f16f8 vs f32 (8 of 64 images): {err:.2e}")
    assert err <= BAR
