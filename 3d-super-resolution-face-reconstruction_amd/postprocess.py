"""Device post-processing behind the sampler: the SR batch goes from the
sampler's output tensor to the MICA / ArcFace encoder inputs without leaving HBM.

Mirrors, per image of the batch, what the reference does on the host one image at a time:
  u8 chain      model/sr3d/model.py:372-386, :462-471 (tensor2img -> cv2.resize 224 -> /255 and
                cv2.dnn.blobFromImages 112, swapRB)
  tensor chain  model/sr3d/model.py:474-483 (tensor2tensor_img * 255 -> create_tensor_blob)
cv2's fixed-point resize is restated from OpenCV's published code (parity unpinned: cv2 is not
installed here); tensor2img and the tensor chain are checked against numpy / torch themselves.
"""
from __future__ import annotations

from typing import Dict

import torch


def _unet(netG_or_unet):
    return getattr(netG_or_unet, "denoise_fn", netG_or_unet)


def _check(sr: torch.Tensor) -> torch.Tensor:
    if sr.dim() == 3:
        sr = sr.unsqueeze(0)
    if sr.dim() != 4 or sr.shape[1] != 3 or not sr.is_cuda:
        raise RuntimeError(f"expected a CUDA tensor [B,3,H,W], got {tuple(sr.shape)} on {sr.device}")
    return sr.to(torch.float32).contiguous()


@torch.no_grad()
def tensor2img(netG_or_unet, sr: torch.Tensor) -> torch.Tensor:
    """Metrics.tensor2img (core/metrics.py:16-42) per image: [B,3,H,W] -> uint8 [B,H,W,3] (RGB)."""
    sr = _check(sr)
    B, _, H, W = sr.shape
    out = torch.empty((B, H, W, 3), dtype=torch.uint8, device=sr.device)
    unet = _unet(netG_or_unet)
    eng = unet.engine()
    unet.ready()
    eng.postprocess_u8(sr.data_ptr(), B, H, W, 0, 0, out.data_ptr(), None, None, None)
    unet.finish()
    return out


@torch.no_grad()
def mica_inputs(netG_or_unet, sr: torch.Tensor, up: int = 224, blob: int = 112) -> Dict[str, torch.Tensor]:
    """The u8 chain: dict(sr_img uint8 [B,H,W,3], sr_up_img uint8 [B,up,up,3], images fp32
    [B,3,up,up] in [0,1], arcface fp32 [B,3,blob,blob] BGR in [-1,1])."""
    sr = _check(sr)
    B, _, H, W = sr.shape
    dev = sr.device
    img = torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev)
    upi = torch.empty((B, up, up, 3), dtype=torch.uint8, device=dev)
    images = torch.empty((B, 3, up, up), dtype=torch.float32, device=dev)
    arc = torch.empty((B, 3, blob, blob), dtype=torch.float32, device=dev)
    unet = _unet(netG_or_unet)
    eng = unet.engine()
    unet.ready()
    eng.postprocess_u8(sr.data_ptr(), B, H, W, up, blob, img.data_ptr(), upi.data_ptr(), images.data_ptr(), arc.data_ptr())
    unet.finish()
    return {"sr_img": img, "sr_up_img": upi, "images": images, "arcface": arc}


@torch.no_grad()
def create_tensor_blob(netG_or_unet, sr: torch.Tensor, blob: int = 112) -> torch.Tensor:
    """The tensor chain (model3): fp32 [B,3,blob,blob]."""
    sr = _check(sr)
    B, _, H, W = sr.shape
    out = torch.empty((B, 3, blob, blob), dtype=torch.float32, device=sr.device)
    unet = _unet(netG_or_unet)
    eng = unet.engine()
    unet.ready()
    eng.postprocess_tensor_blob(sr.data_ptr(), B, H, W, blob, out.data_ptr())
    unet.finish()
    return out
