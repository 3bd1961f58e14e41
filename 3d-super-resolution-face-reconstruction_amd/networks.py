"""`define_G(opt)`: config -> GaussianDiffusion(UNet), as reference model/sr/networks.py:83-116.
`opt` is any mapping with `opt['sr']['model'][...]` and `opt['phase']` (plain dict or yacs node)."""
from __future__ import annotations

from torch import nn

from .diffusion import GaussianDiffusion
from .unet import UNet


def _orthogonal_(net: nn.Module) -> None:
    # networks.py:47-58,110-112: orthogonal weights, zero biases for Conv/Linear in train phase
    for name, p in net.named_parameters():
        if p.dim() >= 2:
            nn.init.orthogonal_(p.data, gain=1)
            bias = dict(net.named_parameters()).get(name.rsplit(".", 1)[0] + ".bias")
            if bias is not None:
                bias.data.zero_()


def define_G(opt):
    model_opt = opt["sr"]["model"]
    which = model_opt["which_model_G"]
    if which != "sr3":
        raise NotImplementedError(f"which_model_G={which!r}: only 'sr3' is built (no reference yml selects 'ddpm')")
    u = model_opt["unet"]
    norm_groups = u["norm_groups"] if ("norm_groups" in u and u["norm_groups"] is not None) else 32
    model = UNet(
        in_channel=u["in_channel"], out_channel=u["out_channel"], norm_groups=norm_groups,
        inner_channel=u["inner_channel"], channel_mults=u["channel_multiplier"],
        attn_res=u["attn_res"], res_blocks=u["res_blocks"], dropout=u["dropout"],
        image_size=model_opt["diffusion"]["image_size"])
    netG = GaussianDiffusion(
        model, image_size=model_opt["diffusion"]["image_size"],
        channels=model_opt["diffusion"]["channels"], loss_type="l1",
        conditional=model_opt["diffusion"]["conditional"],
        schedule_opt=model_opt["beta_schedule"]["train"])
    if opt["phase"] == "train":
        _orthogonal_(netG)
    return netG
