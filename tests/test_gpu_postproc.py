"""Post-processing row through the C-ABI against oracle/post_oracle.py.
tensor2img: bit-exact vs the numpy restatement of core/metrics.py:16-42. Tensor chain: <= 1e-5 vs
the torch CPU calls the reference itself makes. cv2 chain: bit-exact vs the integer restatement of
OpenCV's algorithm — PARITY UNPINNED (cv2 is not installed; no reference fixture), plus a +-1 grey
level bound against an exact float bilinear resize."""
import numpy as np
import pytest

import post_oracle as post
from conftest import pkg

pytestmark = pytest.mark.gpu
synth = pkg("synth")


def _sr_batch(B, r, seed):
    rs = np.random.RandomState(seed)
    x = synth.synth_cond(B, r, max(4, r // 8), seed) + 0.15 * rs.randn(B, 3, r, r).astype(np.float32)
    x[0, :, :2, :] = 1.5            # out-of-range values exercise the clamp
    x[0, :, -2:, :] = -1.5
    x[-1, 0, ::2, ::2] = (rs.randint(0, 256, (r // 2, r // 2)) / 255.0 * 2 - 1 + 1.0 / 255).astype(np.float32)  # .5 ties
    return x.astype(np.float32)


@pytest.mark.parametrize("B,r,up,blob", [(3, 128, 224, 112), (2, 64, 224, 112), (2, 16, 224, 112),
                                         (2, 128, 200, 112), (1, 112, 0, 112), (2, 128, 0, 112)])
def test_u8_chain(B, r, up, blob):
    Engine = pkg("engine").Engine
    e = Engine(synth.tiny_unet_config(), 0)
    x = _sr_batch(B, r, 11 + r)
    got = e.postprocess_np(x, up, blob)
    e.close()
    for b in range(B):
        img = post.tensor2img(x[b])
        np.testing.assert_array_equal(got["img_u8"][b], img)
        if up:
            want_up = post.cv2_resize_linear_u8(img, up, up)
            np.testing.assert_array_equal(got["up_u8"][b], want_up)
            np.testing.assert_array_equal(got["images"][b],
                                          (want_up.astype(np.float64) / 255.0).transpose(2, 0, 1).astype(np.float32))
            # the fixed-point result is a rounding of the exact bilinear value
            exact = post.float_bilinear_u8(img, up, up)
            assert np.abs(want_up.astype(np.float64) - exact).max() <= 1.0
        else:
            want_up = img
        np.testing.assert_array_equal(got["arcface"][b], post.cv2_blob_from_image(want_up, blob))
    assert got["arcface"].min() >= -1.0 and got["arcface"].max() <= 1.0


@pytest.mark.parametrize("B,r,blob", [(3, 128, 112), (2, 64, 112), (2, 16, 112), (1, 224, 112)])
def test_tensor_chain_vs_torch(B, r, blob):
    Engine = pkg("engine").Engine
    e = Engine(synth.tiny_unet_config(), 0)
    x = _sr_batch(B, r, 5 + r)
    got = e.postprocess_np(x, 0, blob)["tensor_arcface"]
    e.close()
    want = post.tensor_blob_torch(x, blob)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5


def test_torch_facade_mica_inputs():
    import torch
    pp = pkg("postprocess")
    cfg = synth.tiny_unet_config()
    unet = pkg().UNet(in_channel=6, out_channel=3, inner_channel=cfg.inner_channel, norm_groups=32,
                      channel_mults=cfg.channel_mults, attn_res=cfg.attn_res, res_blocks=cfg.res_blocks,
                      dropout=0.0, image_size=cfg.image_size).cuda()
    x = _sr_batch(2, 128, 3)
    xt = torch.from_numpy(x).cuda()
    d = pp.mica_inputs(unet, xt)
    ref = post.u8_chain(x[1])
    np.testing.assert_array_equal(d["sr_img"][1].cpu().numpy(), ref["img_u8"])
    np.testing.assert_array_equal(d["sr_up_img"][1].cpu().numpy(), ref["up_u8"])
    np.testing.assert_array_equal(d["images"][1].cpu().numpy(), ref["images"])
    np.testing.assert_array_equal(d["arcface"][1].cpu().numpy(), ref["arcface"])
    assert tuple(d["arcface"].shape) == (2, 3, 112, 112) and d["sr_up_img"].dtype == torch.uint8
    np.testing.assert_array_equal(pp.tensor2img(unet, xt[0]).cpu().numpy()[0], post.tensor2img(x[0]))
    tb = pp.create_tensor_blob(unet, xt).cpu().numpy()
    assert np.abs(tb - post.tensor_blob_torch(x)).max() <= 1e-5
    with pytest.raises(RuntimeError):
        pp.mica_inputs(unet, xt.cpu())
