"""`Engine`: a thin, torch-free Python handle on one `sr3_ctx` (one per process / GPU).

Everything takes raw device addresses (ints) so it can be driven from torch tensors
(`t.data_ptr()`) or from buffers allocated through the library itself (`DeviceBuffer`).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .graph import UNetConfig


def _cfg_struct(cfg: UNetConfig) -> _lib.UnetCfg:
    s = _lib.UnetCfg()
    s.in_channel, s.out_channel = int(cfg.in_channel), int(cfg.out_channel)
    s.inner_channel, s.norm_groups = int(cfg.inner_channel), int(cfg.norm_groups)
    if len(cfg.channel_mults) > _lib.SR3_MAX_MULTS or len(cfg.attn_res) > _lib.SR3_MAX_ATTN_RES:
        raise ValueError("too many channel_mults / attn_res entries")
    s.n_mults = len(cfg.channel_mults)
    for i, m in enumerate(cfg.channel_mults):
        s.channel_mults[i] = int(m)
    s.n_attn_res = len(cfg.attn_res)
    for i, a in enumerate(cfg.attn_res):
        s.attn_res[i] = int(a)
    s.res_blocks, s.image_size = int(cfg.res_blocks), int(cfg.image_size)
    s.dropout = float(cfg.dropout)
    return s


def _host_f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


class DeviceBuffer:
    """fp32 device array owned by the library (for hosts that do not use torch)."""

    def __init__(self, eng: "Engine", n_floats: int):
        self.eng, self.n = eng, int(n_floats)
        p = C.c_void_p()
        _lib.check(eng.lib.sr3_dev_malloc(eng.ctx, self.n * 4, C.byref(p)))
        self.ptr = p.value

    def upload(self, a) -> "DeviceBuffer":
        h = _host_f32(a).ravel()
        assert h.size == self.n, (h.size, self.n)
        _lib.check(self.eng.lib.sr3_memcpy_h2d(self.eng.ctx, self.ptr, h.ctypes.data, h.nbytes))
        return self

    def download(self, shape=None) -> np.ndarray:
        h = np.empty(self.n, dtype=np.float32)
        _lib.check(self.eng.lib.sr3_memcpy_d2h(self.eng.ctx, h.ctypes.data, self.ptr, h.nbytes))
        return h.reshape(shape) if shape is not None else h

    def free(self):
        if self.ptr and self.eng.ctx:
            _lib.check(self.eng.lib.sr3_dev_free(self.eng.ctx, self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    def __init__(self, cfg: UNetConfig, device: int = 0):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = int(device)
        ctx = C.c_void_p()
        _lib.check(self.lib.sr3_create(C.byref(_cfg_struct(cfg)), self.device, C.byref(ctx)))
        self.ctx = ctx
        self.T = 0

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sr3_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffers -------------------------------------------------------------------------
    def buffer(self, n_floats: int) -> DeviceBuffer:
        return DeviceBuffer(self, n_floats)

    def to_device(self, a) -> DeviceBuffer:
        h = _host_f32(a)
        return DeviceBuffer(self, h.size).upload(h)

    def set_stream(self, stream_handle: Optional[int]):
        _lib.check(self.lib.sr3_set_stream(self.ctx, stream_handle or None))

    PRECISIONS = {"f32": 0, "f16x3": 1, "f16f8": 2}

    def conv_f8_supported(self, B: int, H: int, W: int, Cout: int, Cin: int) -> bool:
        """Does the 'f16f8' mode run a 3x3 / stride-1 conv of this shape with fp8 correction products?"""
        return bool(self.lib.sr3_conv_f8_supported(B, H, W, Cout, Cin))

    def set_precision(self, name: str):
        """'f32': exact fp32 MFMA (default). 'f16x3': split-f16 operands, fp32-equivalent accuracy. 'f16f8': f16x3 with
        the correction products of the MFMA-bound convs on the fp8 matrix path (~6e-5 from the reference, faster)."""
        _lib.check(self.lib.sr3_set_precision(self.ctx, self.PRECISIONS[name]))
        self.precision = name

    def set_range_policy(self, strict: bool):
        """split-f16 mode, activation beyond the fp16 range: strict=True fails the call; False (default)
        finishes it in exact f32 and warns (`Sr3RangeWarning`)."""
        _lib.check(self.lib.sr3_set_range_policy(self.ctx, 1 if strict else 0))

    def fallback_calls(self) -> int:
        return int(self.lib.sr3_fallback_calls(self.ctx))

    def replay_calls(self) -> int:
        """Calls finished after replaying work whose in-place split-K wait had given up (SR3_OK_REPLAYED)."""
        return int(self.lib.sr3_replay_calls(self.ctx))

    def synchronize(self):
        _lib.check(self.lib.sr3_synchronize(self.ctx))

    def wait_for_stream(self, stream_handle: Optional[int]):
        """The library's stream waits (on the device) for the work enqueued so far on `stream_handle`."""
        _lib.check(self.lib.sr3_wait_for_stream(self.ctx, stream_handle or None))

    def stream_wait_for_engine(self, stream_handle: Optional[int]):
        """`stream_handle` waits (on the device) for the library work enqueued so far."""
        _lib.check(self.lib.sr3_stream_wait_for_ctx(self.ctx, stream_handle or None))

    def device_bytes(self) -> int:
        return int(self.lib.sr3_device_bytes(self.ctx))

    # ---- weights -------------------------------------------------------------------------
    def param_list(self) -> List[Tuple[str, Tuple[int, ...]]]:
        n = self.lib.sr3_num_params(self.ctx)
        out = []
        name = C.create_string_buffer(192)
        shape = (C.c_int64 * 4)()
        nd = C.c_int()
        for i in range(n):
            _lib.check(self.lib.sr3_param_info(self.ctx, i, name, 192, shape, C.byref(nd)))
            out.append((name.value.decode(), tuple(int(shape[k]) for k in range(nd.value))))
        return out

    def load_weight(self, name: str, host_array) -> None:
        h = _host_f32(host_array)
        shape = (C.c_int64 * max(1, h.ndim))(*h.shape)
        _lib.check(self.lib.sr3_load_weight(self.ctx, name.encode(), h.ctypes.data, shape, h.ndim))

    def load_state_dict(self, sd: Dict[str, np.ndarray], prefix: str = "") -> None:
        """Loads every parameter the library declares from `sd[prefix + name]` (numpy arrays)."""
        for name, _ in self.param_list():
            self.load_weight(name, sd[prefix + name])

    def weights_missing(self) -> int:
        return int(self.lib.sr3_weights_missing(self.ctx))

    # ---- UNet forward ---------------------------------------------------------------------
    def unet_forward(self, x_ptr: int, nl_ptr: int, B: int, H: int, W: int, out_ptr: int) -> None:
        _lib.check(self.lib.sr3_unet_forward(self.ctx, x_ptr, nl_ptr, B, H, W, out_ptr))

    # ---- sampler --------------------------------------------------------------------------
    def set_schedule(self, bufs: Dict[str, np.ndarray]) -> None:
        nl = _host_f32(bufs["noise_level"])
        T = nl.size - 1
        arrs = [nl] + [_host_f32(bufs[k]) for k in (
            "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
            "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2")]
        for a in arrs[1:]:
            assert a.size == T
        _lib.check(self.lib.sr3_set_schedule(self.ctx, T, *[a.ctypes.data for a in arrs]))
        self.T = T

    def num_frames(self) -> int:
        n = self.lib.sr3_num_frames(self.ctx)
        if n < 0:
            _lib.check(n)
        return n

    def max_batch(self, H: int, W: int) -> int:
        """Largest batch one sample / unet_forward call takes at H x W (4 GiB per activation tensor)."""
        n = self.lib.sr3_max_batch(self.ctx, int(H), int(W))
        if n < 0:
            _lib.check(n)
        return n

    def sample(self, cond_ptr: Optional[int], B: int, H: int, W: int, out_ptr: int,
               noise_ptr: Optional[int] = None, seed: int = 0, image_offset: int = 0,
               frames_ptr: Optional[int] = None) -> None:
        _lib.check(self.lib.sr3_sample(self.ctx, cond_ptr or None, B, H, W, noise_ptr or None,
                                       seed, image_offset, out_ptr, frames_ptr or None))

    def sample_begin(self, cond_ptr, B, H, W, init_noise_ptr=None, seed=0, image_offset=0):
        _lib.check(self.lib.sr3_sample_begin(self.ctx, cond_ptr or None, B, H, W,
                                             init_noise_ptr or None, seed, image_offset))

    def sample_step(self, t: int, noise_ptr: Optional[int] = None):
        _lib.check(self.lib.sr3_sample_step(self.ctx, int(t), noise_ptr or None))

    def sample_end(self, out_ptr: int):
        _lib.check(self.lib.sr3_sample_end(self.ctx, out_ptr))

    def range_check(self) -> None:
        """Raises Sr3Error if a split-f16 store left the fp16 range since the last check."""
        _lib.check(self.lib.sr3_range_check(self.ctx))

    def philox_normal(self, seed: int, image: int, draw: int, n: int) -> np.ndarray:
        buf = self.buffer(n)
        _lib.check(self.lib.sr3_philox_normal(self.ctx, seed, image, draw, n, buf.ptr))
        return buf.download()

    def philox_normal_into(self, seed: int, image: int, draw: int, n: int, out_ptr: int) -> None:
        """Device Philox stream of (seed, image, draw) written to a device buffer (stream-ordered)."""
        _lib.check(self.lib.sr3_philox_normal(self.ctx, seed, image, draw, n, out_ptr))

    # ---- pre-processing ---------------------------------------------------------------------
    def preprocess_bicubic(self, in_ptr: int, B: int, Hin: int, Win: int, Hout: int, Wout: int,
                           out_ptr: int, out_u8_ptr: Optional[int] = None) -> None:
        """uint8 HWC device images -> PIL-exact bicubic -> fp32 NCHW [-1,1] (sr3_preprocess_bicubic)."""
        _lib.check(self.lib.sr3_preprocess_bicubic(self.ctx, in_ptr, B, Hin, Win, Hout, Wout, out_ptr,
                                                   out_u8_ptr or None))

    def preprocess_bicubic_np(self, img_u8: np.ndarray, Hout: int, Wout: int):
        """numpy convenience: [B,H,W,3] uint8 -> (fp32 [B,3,Hout,Wout], uint8 [B,Hout,Wout,3])."""
        a = np.ascontiguousarray(img_u8, dtype=np.uint8)
        B, H, W, _ = a.shape
        nin, nout = a.size, B * Hout * Wout * 3
        din, dout, du8 = self.buffer((nin + 3) // 4), self.buffer(nout), self.buffer((nout + 3) // 4)
        _lib.check(self.lib.sr3_memcpy_h2d(self.ctx, din.ptr, a.ctypes.data, nin))
        self.preprocess_bicubic(din.ptr, B, H, W, Hout, Wout, dout.ptr, du8.ptr)
        u8 = np.empty(nout, dtype=np.uint8)
        _lib.check(self.lib.sr3_memcpy_d2h(self.ctx, u8.ctypes.data, du8.ptr, nout))
        return dout.download((B, 3, Hout, Wout)), u8.reshape(B, Hout, Wout, 3)

    # ---- measurement ------------------------------------------------------------------------
    # ---- post-processing ------------------------------------------------
    def postprocess_u8(self, sr_ptr: int, B: int, H: int, W: int, up: int = 224, blob: int = 112,
                       img_u8_ptr: Optional[int] = None, up_u8_ptr: Optional[int] = None,
                       images_ptr: Optional[int] = None, arcface_ptr: Optional[int] = None) -> None:
        _lib.check(self.lib.sr3_postprocess_u8(self.ctx, sr_ptr, B, H, W, up, blob, img_u8_ptr, up_u8_ptr,
                                               images_ptr, arcface_ptr))

    def postprocess_tensor_blob(self, sr_ptr: int, B: int, H: int, W: int, blob: int, arcface_ptr: int) -> None:
        _lib.check(self.lib.sr3_postprocess_tensor_blob(self.ctx, sr_ptr, B, H, W, blob, arcface_ptr))

    def postprocess_np(self, sr: np.ndarray, up: int = 224, blob: int = 112) -> Dict[str, np.ndarray]:
        """Host-array convenience (tests): fp32 [B,3,H,W] -> dict(img_u8, up_u8, images, arcface,
        tensor_arcface)."""
        sr = _host_f32(sr)
        B, _, H, W = sr.shape
        S = up if up else H
        d_in = self.to_device(sr)
        sizes = {"img_u8": B * H * W * 3, "up_u8": B * S * S * 3}
        bufs = {k: self.buffer((n + 3) // 4) for k, n in sizes.items()}
        bufs["images"] = self.buffer(B * 3 * S * S)
        bufs["arcface"] = self.buffer(B * 3 * blob * blob)
        bufs["tensor_arcface"] = self.buffer(B * 3 * blob * blob)
        self.postprocess_u8(d_in.ptr, B, H, W, up, blob, bufs["img_u8"].ptr,
                            bufs["up_u8"].ptr if up else None, bufs["images"].ptr if up else None,
                            bufs["arcface"].ptr)
        self.postprocess_tensor_blob(d_in.ptr, B, H, W, blob, bufs["tensor_arcface"].ptr)
        out = {}
        for k, shape in (("img_u8", (B, H, W, 3)), ("up_u8", (B, S, S, 3))):
            if k == "up_u8" and not up:
                continue
            raw = bufs[k].download()
            out[k] = raw.view(np.uint8)[: sizes[k]].reshape(shape).copy()
        if up:
            out["images"] = bufs["images"].download((B, 3, S, S))
        out["arcface"] = bufs["arcface"].download((B, 3, blob, blob))
        out["tensor_arcface"] = bufs["tensor_arcface"].download((B, 3, blob, blob))
        for b in list(bufs.values()) + [d_in]:
            b.free()
        return out

    def profile_enable(self, on: bool):
        _lib.check(self.lib.sr3_profile_enable(self.ctx, 1 if on else 0))

    def profile_reset(self):
        _lib.check(self.lib.sr3_profile_reset(self.ctx))

    def profile_get(self) -> Dict[str, Dict[str, float]]:
        out = {}
        for i, fam in enumerate(_lib.FAMILIES):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            _lib.check(self.lib.sr3_profile_get(self.ctx, i, C.byref(ms), C.byref(n), C.byref(fl)))
            out[fam] = {"ms": ms.value, "launches": int(n.value), "flops": fl.value}
        return out

    def bench_conv(self, B, H, W, C0, C1, Cout, ks=3, stride=1, up2=0, mode=2, resid=0, chan_bias=0,
                   iters=10) -> float:
        ms, ams = C.c_float(), C.c_float()
        _lib.check(self.lib.sr3_bench_conv(self.ctx, B, H, W, C0, C1, Cout, ks, stride, up2, mode,
                                           resid, chan_bias, iters, C.byref(ms), C.byref(ams)))
        return ms.value, ams.value

    def profile_dump_csv(self, path: str):
        _lib.check(self.lib.sr3_profile_dump_csv(self.ctx, path.encode()))

    # ---- single ops (numpy in / numpy out; NHWC) --------------------------------------------
    def op_conv2d(self, x0, weight, bias=None, x1=None, stride=1, up2=False, gn_scale=None,
                  gn_shift=None, swish=False, chan_bias=None, resid=None) -> np.ndarray:
        x0 = _host_f32(x0)
        B, H, W, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[-1]
        weight = _host_f32(weight)
        Cout, Cin, ks, _ = weight.shape
        assert Cin == C0 + C1
        pad = ks // 2
        Hv, Wv = (H * 2, W * 2) if up2 else (H, W)
        Ho, Wo = (Hv + 2 * pad - ks) // stride + 1, (Wv + 2 * pad - ks) // stride + 1
        d0 = self.to_device(x0)
        d1 = self.to_device(x1) if x1 is not None else None
        dsc = self.to_device(gn_scale) if gn_scale is not None else None
        dsh = self.to_device(gn_shift) if gn_shift is not None else None
        dcb = self.to_device(chan_bias) if chan_bias is not None else None
        drs = self.to_device(resid) if resid is not None else None
        out = self.buffer(B * Ho * Wo * Cout)
        bh = _host_f32(bias) if bias is not None else None
        _lib.check(self.lib.sr3_op_conv2d(
            self.ctx, d0.ptr, C0, d1.ptr if d1 else None, C1, B, H, W, weight.ctypes.data,
            bh.ctypes.data if bh is not None else None, Cout, ks, stride, 1 if up2 else 0,
            dsc.ptr if dsc else None, dsh.ptr if dsh else None, 1 if swish else 0,
            dcb.ptr if dcb else None, drs.ptr if drs else None, out.ptr))
        return out.download((B, Ho, Wo, Cout))

    def op_groupnorm_affine(self, x0, gamma, beta, groups=32, x1=None):
        x0 = _host_f32(x0)
        B, H, W, C0 = x0.shape
        C1 = 0 if x1 is None else x1.shape[-1]
        C = C0 + C1
        d0 = self.to_device(x0)
        d1 = self.to_device(x1) if x1 is not None else None
        sc, sh = self.buffer(B * C), self.buffer(B * C)
        g, b = _host_f32(gamma), _host_f32(beta)
        _lib.check(self.lib.sr3_op_groupnorm_affine(
            self.ctx, d0.ptr, C0, d1.ptr if d1 else None, C1, B, H, W, groups, g.ctypes.data,
            b.ctypes.data, sc.ptr, sh.ptr))
        return sc.download((B, C)), sh.download((B, C))

    def op_attention(self, qkv) -> np.ndarray:
        qkv = _host_f32(qkv)
        B, N, C3 = qkv.shape
        d = self.to_device(qkv)
        out = self.buffer(B * N * (C3 // 3))
        _lib.check(self.lib.sr3_op_attention(self.ctx, d.ptr, B, N, C3 // 3, out.ptr))
        return out.download((B, N, C3 // 3))

    def op_noise_embed(self, noise_level) -> Tuple[np.ndarray, np.ndarray]:
        nl = _host_f32(noise_level).ravel()
        B = nl.size
        total = self.lib.sr3_chan_bias_total(self.ctx)
        d = self.to_device(nl)
        te, cb = self.buffer(B * self.cfg.inner_channel), self.buffer(B * total)
        _lib.check(self.lib.sr3_op_noise_embed(self.ctx, d.ptr, B, te.ptr, cb.ptr))
        return te.download((B, self.cfg.inner_channel)), cb.download((B, total))

    # ---- numpy conveniences (NCHW in / out) ---------------------------------------------------
    def unet_forward_np(self, x, noise_level) -> np.ndarray:
        x = _host_f32(x)
        B, _, H, W = x.shape
        dx, dn = self.to_device(x), self.to_device(np.asarray(noise_level).ravel())
        out = self.buffer(B * self.cfg.out_channel * H * W)
        self.unet_forward(dx.ptr, dn.ptr, B, H, W, out.ptr)
        return out.download((B, self.cfg.out_channel, H, W))

    def sample_np(self, cond, noise=None, seed=0, image_offset=0, frames=False, shape=None):
        """cond [B,Cc,H,W] (or None with shape=(B,C,H,W)); noise [T,B,C,H,W] or None (Philox)."""
        C_ = self.cfg.out_channel
        if cond is not None:
            cond = _host_f32(cond)
            B, _, H, W = cond.shape
            dc = self.to_device(cond)
        else:
            B, _, H, W = shape
            dc = None
        dn = self.to_device(noise) if noise is not None else None
        out = self.buffer(B * C_ * H * W)
        nf = self.num_frames()
        fr = self.buffer(nf * B * C_ * H * W) if frames else None
        self.sample(dc.ptr if dc else None, B, H, W, out.ptr, dn.ptr if dn else None, seed,
                    image_offset, fr.ptr if fr else None)
        o = out.download((B, C_, H, W))
        if frames:
            return o, fr.download((nf, B, C_, H, W))
        return o
