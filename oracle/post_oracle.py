"""TEST INFRASTRUCTURE ONLY (see oracle/README.md) — CPU restatement of the reference's host-side
post-processing between the sampler and the MICA / ArcFace encoder.

Pinned parts:
  * tensor2img (core/metrics.py:16-42) is plain numpy/torch arithmetic, restated 1:1;
  * the tensor chain tensor2tensor_img * 255 -> create_tensor_blob (core/metrics.py:44-50,
    model/sr3d/model.py:105-124) is evaluated with the very torch functions the reference calls
    (torch.clamp, F.interpolate(mode='bilinear', align_corners=False)) on the CPU.
PARITY UNPINNED: cv2 (opencv-python) is a third-party dependency of the reference that is not
installed in this image, and the reference holds no fixture for it. `cv2_resize_linear_u8` and
`cv2_blob_from_image` restate the published OpenCV 4.x algorithm (modules/imgproc/src/resize.cpp:
coefficient set-up in resize(), HResizeLinear<uchar,int,short,INTER_RESIZE_COEF_SCALE>,
VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>, the "INTER_LINEAR with scale 2 ==
INTER_AREA" shortcut and ResizeAreaFastVec; modules/dnn/src/dnn_utils.cpp: blobFromImages) in
plain integer numpy, independently of the HIP kernel (vectorised here, per-thread there). As an
additional sanity check the fixed-point result must stay within 1 grey level of an exact float
bilinear evaluation (tests/test_gpu_postproc.py).
"""
import numpy as np

COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS


def tensor2img(x_chw: np.ndarray) -> np.ndarray:
    """core/metrics.py:16-42 for one image: [3,H,W] fp32 -> [H,W,3] uint8 (RGB)."""
    t = np.clip(x_chw.astype(np.float32), np.float32(-1), np.float32(1))
    t = (t - np.float32(-1)) / np.float32(2)
    img = np.transpose(t, (1, 2, 0))
    img = (img * np.float32(255.0)).round()            # numpy: half to even, float32
    return img.astype(np.uint8)


def _linear_coeffs(in_size: int, out_size: int, horizontal: bool):
    scale = 1.0 / (float(out_size) / float(in_size))
    ofs = np.zeros(out_size, np.int64)
    ab = np.zeros((out_size, 2), np.int64)
    for d in range(out_size):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - np.float32(s))
        if horizontal:
            if s < 0:
                f, s = np.float32(0), 0
            if s >= in_size - 1:
                f, s = np.float32(0), in_size - 1
        ofs[d] = s
        c = (np.float32(1) - f, f)
        for k in range(2):
            ab[d, k] = int(np.rint(np.float32(c[k] * np.float32(COEF_ONE))))
    return ofs, ab


def cv2_resize_linear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h)) for uint8 HWC input, default INTER_LINEAR."""
    H, W, _ = img.shape
    xo, xa = _linear_coeffs(W, out_w, True)
    yo, yb = _linear_coeffs(H, out_h, False)
    src = img.astype(np.int64)
    x1 = np.minimum(xo + 1, W - 1)
    # horizontal pass on every source row: [H, out_w, C], 11 fractional bits
    rows = src[:, xo, :] * xa[None, :, 0, None] + src[:, x1, :] * xa[None, :, 1, None]
    y0 = np.clip(yo, 0, H - 1)
    y1 = np.clip(yo + 1, 0, H - 1)
    S0, S1 = rows[y0], rows[y1]
    b0, b1 = yb[:, 0, None, None], yb[:, 1, None, None]
    v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2
    return (v & 0xFF).astype(np.uint8)


def cv2_blob_from_image(img: np.ndarray, size: int, mean: float = 127.5, std: float = 127.5) -> np.ndarray:
    """cv2.dnn.blobFromImages([img], 1/std, (size, size), (mean,)*3, swapRB=True)[0]:
    [3,size,size] fp32, channel order reversed."""
    H, W, _ = img.shape
    if (H, W) != (size, size):
        if H == 2 * size and W == 2 * size:         # INTER_LINEAR at scale 2 -> INTER_AREA fast path
            s = img.astype(np.int64)
            img = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        else:
            img = cv2_resize_linear_u8(img, size, size)
    f = img.astype(np.float32)
    f = (f - np.float32(mean)) * np.float32(1.0 / std)
    return np.ascontiguousarray(np.transpose(f[:, :, ::-1], (2, 0, 1)))


def u8_chain(x_chw: np.ndarray, up: int = 224, blob: int = 112):
    """model/sr3d/model.py:372-386 for one image -> dict(img_u8, up_u8, images, arcface)."""
    img = tensor2img(x_chw)
    up_u8 = cv2_resize_linear_u8(img, up, up) if up else img
    images = (up_u8.astype(np.float64) / 255.0).transpose(2, 0, 1).astype(np.float32)
    return {"img_u8": img, "up_u8": up_u8, "images": images, "arcface": cv2_blob_from_image(up_u8, blob)}


def float_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """Exact (float64) half-pixel-centre bilinear resize with edge replication: the function cv2's
    fixed-point code approximates; used only as a +-1 sanity bound."""
    H, W, _ = img.shape
    ys = np.clip((np.arange(out_h) + 0.5) * (H / out_h) - 0.5, 0, H - 1)
    xs = np.clip((np.arange(out_w) + 0.5) * (W / out_w) - 0.5, 0, W - 1)
    y0 = np.floor(ys).astype(int); x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    fy = (ys - y0)[:, None, None]; fx = (xs - x0)[None, :, None]
    s = img.astype(np.float64)
    top = s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def tensor_blob_torch(x_bchw: np.ndarray, blob: int = 112) -> np.ndarray:
    """model/sr3d/model.py:474-483 per image, with the torch calls of core/metrics.py:44-50 and
    model/sr3d/model.py:105-124 themselves (CPU)."""
    import torch
    import torch.nn.functional as F
    outs = []
    for x in torch.from_numpy(np.ascontiguousarray(x_bchw, dtype=np.float32)):
        t = x.float().cpu().clamp_(-1, 1)
        t = (t - (-1)) / (1 - (-1))
        t = t * 255.0
        t = (t - 127.5) / 127.5
        r = F.interpolate(t.unsqueeze(0), size=(blob, blob), mode="bilinear", align_corners=False).squeeze(0)
        outs.append(r[[2, 1, 0], :, :].numpy())
    return np.stack(outs)
