"""Single-kernel parity on the GPU: each HIP op called through the C-ABI (sr3_op_*) against the
oracle's restatement of the same reference op. fp32; tolerances are absolute on O(1) data."""
import numpy as np
import pytest

import sr3_oracle as oracle
from conftest import pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")


@pytest.fixture(scope="module")
def eng():
    Engine = pkg("engine").Engine
    e = Engine(synth.tiny_unet_config(), 0)
    e.load_state_dict(synth.synth_state_dict(e.cfg, 11))
    yield e
    e.close()


def _rand(rs, *shape):
    return rs.standard_normal(shape).astype(np.float32)


CONV_CASES = [
    # B, H, W, C0, C1, Cout, ks, stride, up2
    (2, 16, 16, 32, 0, 64, 3, 1, False),
    (2, 16, 16, 64, 32, 64, 3, 1, False),     # concat input, 64x64 tile path
    (1, 32, 32, 64, 0, 128, 3, 1, False),     # 128-wide N
    (3, 8, 8, 128, 0, 128, 3, 2, False),      # Downsample (unet.py:68-74)
    (2, 8, 8, 64, 0, 64, 3, 1, True),         # Upsample (unet.py:58-65)
    (2, 12, 20, 96, 0, 32, 1, 1, False),      # 1x1, non-square, ragged M tile
    (1, 16, 16, 32, 0, 3, 3, 1, False),       # Cout = 3 (final_conv)
    (5, 4, 4, 256, 256, 256, 1, 1, False),    # res_conv over a concatenation
    (64, 8, 8, 64, 0, 128, 3, 1, False),      # enough blocks for the 128x128 tile
    (40, 16, 16, 64, 0, 64, 3, 1, False),     # 128x64 tile
    # shapes large enough for the x-halo kernels (f16x3): M >= 65536 pixels
    (4, 128, 128, 64, 0, 64, 3, 1, False),    # 128x64 halo tile, deep B ring, one row segment per tile
    (16, 64, 64, 64, 0, 64, 3, 1, False),     # two row segments per tile
    (64, 32, 32, 64, 32, 64, 3, 1, False),    # four segments, concatenated input (chunk 2 comes from in1)
    (16, 64, 64, 64, 0, 64, 3, 1, True),      # Upsample as four 2x2 phase convs on the halo kernel
    (64, 32, 32, 128, 0, 128, 3, 1, False),   # 128x128 halo tile
    (64, 8, 8, 512, 0, 512, 3, 1, False),     # small M, deep K (the 8x8 level): 64x64 tile, 4-stage ring
    (64, 16, 16, 256, 0, 256, 3, 2, False),   # Downsample at full batch
    # few 128x128 tiles, deep K: x-halo tile with in-place split-K (f16x3; conv_halo_splits)
    (64, 8, 8, 512, 512, 512, 3, 1, False),   # tiles cover two whole images each, concatenated input
    (32, 8, 8, 512, 0, 512, 3, 1, False),     # 64 tiles
    (36, 8, 8, 512, 0, 512, 3, 1, False),     # 72 tiles: not a multiple of 8 (padded block order)
    (16, 16, 16, 512, 0, 256, 3, 1, False),   # two tiles per image, two N-tiles
    # split-K shapes (few tiles, deep K: small batches)
    (1, 8, 8, 512, 0, 512, 3, 1, False),      # one M-tile, 8 N-tiles: in-place split-K (whole tiles per image)
    (1, 16, 16, 256, 256, 256, 3, 1, False),  # in place, concatenated input, 4 M-tiles
    (4, 4, 4, 256, 0, 256, 3, 1, False),      # a tile spans four images: conv + reduce kernel
    (2, 2, 2, 512, 0, 512, 3, 1, False),      # ragged M tile (8 pixels): conv + reduce kernel
    (1, 8, 8, 256, 0, 256, 3, 1, True),       # Upsample phases with split-K (4 phases x in-place tiles)
    (3, 8, 8, 256, 0, 768, 1, 1, False),      # 1x1 (attention qkv) with split-K, three M-tiles
]


PRECISIONS = ["f32", "f16x3"]   # exact fp32 MFMA | split-f16 (hi+lo operands, fp32 accumulate)


@pytest.mark.parametrize("prec", PRECISIONS)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d(eng, case, prec):
    eng.set_precision(prec)
    B, H, W, C0, C1, Cout, ks, stride, up2 = case
    rs = np.random.RandomState(hash(case) & 0xFFFF)
    x0 = _rand(rs, B, H, W, C0)
    x1 = _rand(rs, B, H, W, C1) if C1 else None
    w = _rand(rs, Cout, C0 + C1, ks, ks) / np.sqrt((C0 + C1) * ks * ks)
    b = _rand(rs, Cout)
    got = eng.op_conv2d(x0, w, b, x1=x1, stride=stride, up2=up2)
    xin = x0 if x1 is None else np.concatenate([x0, x1], -1)
    if up2:
        xin = oracle.upsample_nearest2(xin)
    want = oracle.conv2d(xin, w, b, stride=stride)
    eng.set_precision("f32")
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)


@pytest.mark.parametrize("prec", PRECISIONS)
def test_conv2d_fused_prologue_epilogue(eng, prec):
    """GroupNorm apply + Swish pass, conv, FeatureWiseAffine bias and residual in the epilogue ==
    Block + noise_func + residual of ResnetBlock.forward (unet.py:105-110)."""
    eng.set_precision(prec)
    rs = np.random.RandomState(5)
    B, H, W, C, Cout = 3, 16, 16, 64, 64
    x = _rand(rs, B, H, W, C) * 2 + 0.5
    gamma, beta = 1 + 0.1 * _rand(rs, C), 0.1 * _rand(rs, C)
    w, b = _rand(rs, Cout, C, 3, 3) / 24, _rand(rs, Cout)
    cb, resid = _rand(rs, B, Cout), _rand(rs, B, H, W, Cout)
    sc, sh = eng.op_groupnorm_affine(x, gamma, beta, 32)
    got = eng.op_conv2d(x, w, b, gn_scale=sc, gn_shift=sh, swish=True, chan_bias=cb, resid=resid)
    want = oracle.conv2d(oracle.swish(oracle.group_norm(x, gamma, beta, 32)), w, b)
    want = want + cb[:, None, None, :] + resid
    eng.set_precision("f32")
    np.testing.assert_allclose(got, want, atol=3e-5, rtol=0)


def test_conv2d_f16x3_wide_dynamic_range(eng):
    """split-f16 keeps fp32-level accuracy for tiny and large operands (weights are pre-scaled by
    2^k, activations are bounded by GroupNorm in the network; here up to +-3e3)."""
    rs = np.random.RandomState(9)
    x = (_rand(rs, 2, 16, 16, 64) * np.exp(rs.uniform(-12, 8, (2, 16, 16, 64)))).astype(np.float32)
    w = (_rand(rs, 64, 64, 3, 3) * 1e-3).astype(np.float32)
    eng.set_precision("f16x3")
    got = eng.op_conv2d(x, w, None)
    eng.set_precision("f32")
    want = oracle.conv2d(x.astype(np.float64), w.astype(np.float64))
    scale = np.abs(want).max()
    assert np.abs(got - want).max() < 2e-6 * scale


@pytest.mark.parametrize("shape", [(2, 16, 16, 32, 0), (2, 16, 16, 64, 32), (3, 8, 8, 512, 256),
                                   (1, 4, 4, 1024, 0), (2, 64, 64, 64, 0), (2, 1, 1, 512, 0),
                                   (70, 8, 8, 128, 64)])
def test_groupnorm_affine(eng, shape):
    B, H, W, C0, C1 = shape
    rs = np.random.RandomState(sum(shape))
    x0 = _rand(rs, B, H, W, C0) * 3 + 10.0          # large mean: exercises the Welford/Chan path
    x1 = (_rand(rs, B, H, W, C1) - 4.0) if C1 else None
    C = C0 + C1
    gamma, beta = 1 + 0.1 * _rand(rs, C), 0.1 * _rand(rs, C)
    sc, sh = eng.op_groupnorm_affine(x0, gamma, beta, 32, x1=x1)
    x = x0 if x1 is None else np.concatenate([x0, x1], -1)
    got = x * sc[:, None, None, :] + sh[:, None, None, :]
    want = oracle.group_norm(x, gamma, beta, 32)
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("B,N,C", [(2, 64, 512), (1, 256, 512), (3, 1, 64), (2, 4, 32), (2, 100, 64), (1, 16, 512),
                                   (3, 256, 64), (1, 1024, 32), (64, 256, 64), (128, 128, 96)])
def test_attention(eng, B, N, C, prec):
    """f32: f32-MFMA core; f16x3: the split-f16 core (q, k, v and the probabilities as hi + lo halfs); the last two
    shapes have 256+ blocks of 64 queries: the 64-query x 8-wave form of the split core (BASELINE config 3 at B = 64)."""
    rs = np.random.RandomState(B * 1000 + N + C)
    qkv = _rand(rs, B, N, 3 * C)
    eng.set_precision(prec)
    try:
        got = eng.op_attention(qkv)
    finally:
        eng.set_precision("f32")
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    s = np.einsum("bpc,bqc->bpq", q, k).astype(np.float64) / np.sqrt(C)
    s = np.exp(s - s.max(-1, keepdims=True))
    want = np.einsum("bpq,bqc->bpc", s / s.sum(-1, keepdims=True), v)
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)


def test_attention_peaked_softmax(eng):
    """Scores far apart (one key dominates) and a row of identical scores."""
    B, N, C = 1, 64, 64
    rs = np.random.RandomState(3)
    qkv = _rand(rs, B, N, 3 * C)
    qkv[0, 5, :C] *= 40.0
    qkv[0, 7, :C] = 0.0
    got = eng.op_attention(qkv)
    eng.set_precision("f16x3")
    try:
        got16 = eng.op_attention(qkv)
    finally:
        eng.set_precision("f32")
    assert np.abs(got16 - got).max() < 5e-5
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    s = np.einsum("bpc,bqc->bpq", q, k).astype(np.float64) / np.sqrt(C)
    s = np.exp(s - s.max(-1, keepdims=True))
    want = np.einsum("bpq,bqc->bpc", s / s.sum(-1, keepdims=True), v)
    np.testing.assert_allclose(got, want, atol=5e-5, rtol=0)


def test_noise_embed(eng):
    cfg = eng.cfg
    sd = synth.synth_state_dict(cfg, 11)
    nl = np.array([0.0, 0.013, 0.5, 0.99999, 1.0], dtype=np.float32)
    temb, cb = eng.op_noise_embed(nl)
    want = oracle.noise_level_mlp(sd, "", nl, cfg.inner_channel)
    np.testing.assert_allclose(temb, want, atol=2e-6, rtol=0)
    off = 0
    for name in sorted((k for k in sd if k.endswith("noise_func.noise_func.0.weight")),
                       key=lambda k: [n for n, _, _ in pkg("graph").param_specs(cfg)].index(k)):
        w, b = sd[name], sd[name.replace("weight", "bias")]
        np.testing.assert_allclose(cb[:, off:off + w.shape[0]], want @ w.T + b, atol=5e-6, rtol=0)
        off += w.shape[0]
    assert off == cb.shape[1]


def test_philox_twin(eng):
    import philox
    for seed, image, draw in [(0, 0, 0), (12345678901234567, 3, 7), (2 ** 63 + 5, 2 ** 33 + 1, 999)]:
        got = eng.philox_normal(seed, image, draw, 3 * 16 * 16 + 3)
        want = philox.normal(seed, image, draw, got.size)
        np.testing.assert_allclose(got, want, atol=2e-5, rtol=0)
    z = eng.philox_normal(1, 0, 1, 1 << 16)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
