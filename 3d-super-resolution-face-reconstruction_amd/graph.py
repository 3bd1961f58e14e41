"""Host-side mirror of the UNet parameter inventory.

`param_specs(cfg)` lists (name, shape, kind) in the registration order of the reference's
`UNet.__init__` (model/sr/sr3_modules/unet.py:161-233), i.e. the `denoise_fn.*` part of the
state_dict described in SURVEY.md §8a. The HIP library builds the same list in C++
(csrc/sr3_api.hip build_graph); tests compare the two and the reference's own key list.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Iterable, List, Sequence, Tuple


@dataclass
class UNetConfig:
    in_channel: int = 6
    out_channel: int = 3
    inner_channel: int = 32
    norm_groups: int = 32
    channel_mults: Sequence[int] = (1, 2, 4, 8, 8)
    attn_res: Sequence[int] = (8,)
    res_blocks: int = 3
    dropout: float = 0.0
    image_size: int = 128

    def __post_init__(self):
        ar = self.attn_res
        # the reference default is the int `(8)`; `x in (8)` would raise there, lists are what yml gives
        self.attn_res = tuple(ar) if isinstance(ar, Iterable) else (int(ar),)
        self.channel_mults = tuple(int(m) for m in self.channel_mults)
        if self.norm_groups is None:
            self.norm_groups = 32


ParamSpec = Tuple[str, Tuple[int, ...], str]   # kind: conv | linear | norm_w | norm_b | bias


def _res(out: List[ParamSpec], prefix: str, cin: int, cout: int, inner: int, attn: bool) -> None:
    rp = prefix + ".res_block"
    out.append((rp + ".noise_func.noise_func.0.weight", (cout, inner), "linear"))
    out.append((rp + ".noise_func.noise_func.0.bias", (cout,), "bias"))
    out.append((rp + ".block1.block.0.weight", (cin,), "norm_w"))
    out.append((rp + ".block1.block.0.bias", (cin,), "norm_b"))
    out.append((rp + ".block1.block.3.weight", (cout, cin, 3, 3), "conv"))
    out.append((rp + ".block1.block.3.bias", (cout,), "bias"))
    out.append((rp + ".block2.block.0.weight", (cout,), "norm_w"))
    out.append((rp + ".block2.block.0.bias", (cout,), "norm_b"))
    out.append((rp + ".block2.block.3.weight", (cout, cout, 3, 3), "conv"))
    out.append((rp + ".block2.block.3.bias", (cout,), "bias"))
    if cin != cout:
        out.append((rp + ".res_conv.weight", (cout, cin, 1, 1), "conv"))
        out.append((rp + ".res_conv.bias", (cout,), "bias"))
    if attn:
        out.append((prefix + ".attn.norm.weight", (cout,), "norm_w"))
        out.append((prefix + ".attn.norm.bias", (cout,), "norm_b"))
        out.append((prefix + ".attn.qkv.weight", (3 * cout, cout, 1, 1), "conv"))
        out.append((prefix + ".attn.out.weight", (cout, cout, 1, 1), "conv"))
        out.append((prefix + ".attn.out.bias", (cout,), "bias"))


def param_specs(cfg: UNetConfig) -> List[ParamSpec]:
    inner = cfg.inner_channel
    out: List[ParamSpec] = [
        ("noise_level_mlp.1.weight", (4 * inner, inner), "linear"),
        ("noise_level_mlp.1.bias", (4 * inner,), "bias"),
        ("noise_level_mlp.3.weight", (inner, 4 * inner), "linear"),
        ("noise_level_mlp.3.bias", (inner,), "bias"),
        ("downs.0.weight", (inner, cfg.in_channel, 3, 3), "conv"),
        ("downs.0.bias", (inner,), "bias"),
    ]
    pre, now_res, idx = inner, cfg.image_size, 1
    feat = [pre]
    n = len(cfg.channel_mults)
    for ind, mult in enumerate(cfg.channel_mults):
        attn = now_res in cfg.attn_res
        ch = inner * mult
        for _ in range(cfg.res_blocks):
            _res(out, f"downs.{idx}", pre, ch, inner, attn)
            idx += 1
            feat.append(ch)
            pre = ch
        if ind != n - 1:
            out.append((f"downs.{idx}.conv.weight", (pre, pre, 3, 3), "conv"))
            out.append((f"downs.{idx}.conv.bias", (pre,), "bias"))
            idx += 1
            feat.append(pre)
            now_res //= 2
    _res(out, "mid.0", pre, pre, inner, True)
    _res(out, "mid.1", pre, pre, inner, False)
    idx = 0
    for ind in reversed(range(n)):
        attn = now_res in cfg.attn_res
        ch = inner * cfg.channel_mults[ind]
        for _ in range(cfg.res_blocks + 1):
            _res(out, f"ups.{idx}", pre + feat.pop(), ch, inner, attn)
            idx += 1
            pre = ch
        if ind >= 1:
            out.append((f"ups.{idx}.conv.weight", (pre, pre, 3, 3), "conv"))
            out.append((f"ups.{idx}.conv.bias", (pre,), "bias"))
            idx += 1
            now_res *= 2
    out.append(("final_conv.block.0.weight", (pre,), "norm_w"))
    out.append(("final_conv.block.0.bias", (pre,), "norm_b"))
    out.append(("final_conv.block.3.weight", (cfg.out_channel, pre, 3, 3), "conv"))
    out.append(("final_conv.block.3.bias", (cfg.out_channel,), "bias"))
    return out


def count_params(cfg: UNetConfig) -> int:
    total = 0
    for _, shape, _ in param_specs(cfg):
        n = 1
        for d in shape:
            n *= d
        total += n
    return total


def flops_per_image(cfg: UNetConfig, h: int, w: int) -> float:
    """Algorithmic FLOPs (2*MAC of Conv2d + Linear + QK^T + PV) of one UNet forward for one image
    of size h x w — the work unit of SURVEY.md §8d (89.00 GFLOP at 128x128 for the yml-literal
    config)."""
    inner = cfg.inner_channel
    fl = 0.0

    def conv(cin, cout, k, hh, ww):
        return 2.0 * hh * ww * cout * cin * k * k

    def res(cin, cout, hh, ww, attn):
        f = 2.0 * inner * cout                       # FeatureWiseAffine linear
        f += conv(cin, cout, 3, hh, ww) + conv(cout, cout, 3, hh, ww)
        if cin != cout:
            f += conv(cin, cout, 1, hh, ww)
        if attn:
            n = hh * ww
            f += conv(cout, 3 * cout, 1, hh, ww) + conv(cout, cout, 1, hh, ww) + 4.0 * n * n * cout
        return f

    fl += 2.0 * inner * 4 * inner * 2                # noise_level_mlp
    fl += conv(cfg.in_channel, inner, 3, h, w)
    pre, now_res = inner, cfg.image_size
    feat = [pre]
    n = len(cfg.channel_mults)
    for ind, mult in enumerate(cfg.channel_mults):
        attn = now_res in cfg.attn_res
        ch = inner * mult
        for _ in range(cfg.res_blocks):
            fl += res(pre, ch, h, w, attn)
            feat.append(ch)
            pre = ch
        if ind != n - 1:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            fl += conv(pre, pre, 3, h, w)
            feat.append(pre)
            now_res //= 2
    fl += res(pre, pre, h, w, True) + res(pre, pre, h, w, False)
    for ind in reversed(range(n)):
        attn = now_res in cfg.attn_res
        ch = inner * cfg.channel_mults[ind]
        for _ in range(cfg.res_blocks + 1):
            fl += res(pre + feat.pop(), ch, h, w, attn)
            pre = ch
        if ind >= 1:
            h, w = 2 * h, 2 * w
            fl += conv(pre, pre, 3, h, w)
            now_res *= 2
    fl += conv(pre, cfg.out_channel, 3, h, w)
    return fl


def upsample_flops_per_image(cfg: UNetConfig, h: int, w: int) -> float:
    """Algorithmic FLOPs of the `Upsample` convs alone (unet.py:58-65; part of flops_per_image).
    The HIP path runs them as four sub-pixel 2x2 convs = 16/36 of these MACs, which the roofline's
    `executed_frac` accounts for."""
    inner, n = cfg.inner_channel, len(cfg.channel_mults)
    div = 2 ** (n - 1)
    h, w = h // div, w // div
    fl = 0.0
    for ind in reversed(range(1, n)):
        ch = inner * cfg.channel_mults[ind]
        h, w = 2 * h, 2 * w
        fl += 2.0 * h * w * ch * ch * 9
    return fl


def resblock_conv3x3_shapes(cfg: UNetConfig, h: int, w: int):
    """(h, w, cin, cout) of the two 3x3 convs of every ResnetBlock of one forward at h x w (unet.py:94-110, 161-233) —
    the convs the f16f8 arithmetic may run with fp8 correction products (Engine.conv_f8_supported decides per shape)."""
    inner, n = cfg.inner_channel, len(cfg.channel_mults)
    out = []

    def res(cin, cout, hh, ww):
        out.append((hh, ww, cin, cout))
        out.append((hh, ww, cout, cout))

    pre = inner
    feat = [pre]
    for ind, mult in enumerate(cfg.channel_mults):
        ch = inner * mult
        for _ in range(cfg.res_blocks):
            res(pre, ch, h, w)
            feat.append(ch)
            pre = ch
        if ind != n - 1:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            feat.append(pre)
    res(pre, pre, h, w)
    res(pre, pre, h, w)
    for ind in reversed(range(n)):
        ch = inner * cfg.channel_mults[ind]
        for _ in range(cfg.res_blocks + 1):
            res(pre + feat.pop(), ch, h, w)
            pre = ch
        if ind >= 1:
            h, w = 2 * h, 2 * w
    return out

