"""Device pre-processing in front of the sampler: raw l x l uint8 crops ->
the `SR` conditioning tensor the reference's dataset would have produced with PIL
(datasets/tool/prepare_data.py:24-47 + datasets/util.py:76-83), bit-exactly (Pillow 12.2)."""
from __future__ import annotations

import torch


@torch.no_grad()
def bicubic_sr(netG_or_unet, lr_u8: torch.Tensor, r: int, return_u8: bool = False):
    """lr_u8: [B,l,l,3] uint8 CUDA tensor (HWC, RGB) -> fp32 [B,3,r,r] in [-1,1] on the same device
    (and the resized uint8 [B,r,r,3] if return_u8)."""
    unet = getattr(netG_or_unet, "denoise_fn", netG_or_unet)
    eng = unet.engine()
    if lr_u8.dtype != torch.uint8 or lr_u8.dim() != 4 or lr_u8.shape[-1] != 3:
        raise RuntimeError(f"expected uint8 [B,H,W,3], got {lr_u8.dtype} {tuple(lr_u8.shape)}")
    x = lr_u8.contiguous()
    B, H, W, _ = x.shape
    out = torch.empty((B, 3, r, r), dtype=torch.float32, device=x.device)
    u8 = torch.empty((B, r, r, 3), dtype=torch.uint8, device=x.device) if return_u8 else None
    unet.ready()
    eng.preprocess_bicubic(x.data_ptr(), B, H, W, r, r, out.data_ptr(), u8.data_ptr() if return_u8 else None)
    unet.finish()
    return (out, u8) if return_u8 else out
