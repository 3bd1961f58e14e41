"""`UNet`: the reference's denoiser API (model/sr/sr3_modules/unet.py:161-265) over libsr3hip.

Same constructor keywords, same `forward(x, time)` contract, same `state_dict` keys and tensor
layouts (Conv2d OIHW, Linear [out,in]) — so `load_state_dict` of a reference checkpoint works —
but no torch compute: parameters are only *storage*; `forward` hands device pointers to the HIP
library, which keeps its own kernel-layout copy of the weights. There is no CPU path.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
from torch import nn

from ._lib import Sr3Error
from .engine import Engine
from .graph import UNetConfig, param_specs


class _Node(nn.Module):
    """Parameter container; only gives the reference's dotted names to the parameters."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("container module; compute happens in UNet.forward (HIP)")

    def __getitem__(self, idx):
        # numeric children behave like the reference's ModuleList / Sequential entries
        return self._modules[str(idx)]

    def __len__(self):
        return len(self._modules)


def _register(root: nn.Module, dotted: str, p: nn.Parameter) -> None:
    parts = dotted.split(".")
    mod = root
    for name in parts[:-1]:
        nxt = mod._modules.get(name)
        if nxt is None:
            nxt = _Node()
            mod.add_module(name, nxt)
        mod = nxt
    mod.register_parameter(parts[-1], p)


def _default_init(shape, kind: str) -> torch.Tensor:
    # PyTorch's default resets of Conv2d / Linear / GroupNorm, which the reference relies on
    # when phase != 'train' (networks.py:110-112)
    t = torch.empty(shape, dtype=torch.float32)
    if kind in ("conv", "linear"):
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
    elif kind == "norm_w":
        t.fill_(1.0)
    elif kind == "norm_b":
        t.zero_()
    else:
        t.zero_()
    return t


class UNet(nn.Module):
    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), res_blocks=3, dropout=0,
                 with_noise_level_emb=True, image_size=128):
        super().__init__()
        if not with_noise_level_emb:
            raise NotImplementedError("with_noise_level_emb=False is not used by any reference config")
        self.cfg = UNetConfig(in_channel=in_channel, out_channel=out_channel if out_channel else in_channel,
                              inner_channel=inner_channel, norm_groups=norm_groups,
                              channel_mults=channel_mults, attn_res=attn_res, res_blocks=res_blocks,
                              dropout=dropout, image_size=image_size)
        fan_in = {}
        for name, shape, kind in param_specs(self.cfg):
            t = _default_init(shape, kind)
            if kind in ("conv", "linear"):
                fan_in[name.rsplit(".", 1)[0]] = int(torch.tensor(shape[1:]).prod())
            elif kind == "bias":
                fi = fan_in.get(name.rsplit(".", 1)[0])
                if fi:
                    nn.init.uniform_(t, -1.0 / math.sqrt(fi), 1.0 / math.sqrt(fi))
            _register(self, name, nn.Parameter(t))
        self._engine: Optional[Engine] = None
        self._synced = {}
        # conv arithmetic: "f16x3" = split-f16 operands with fp32 accumulation (fp32-equivalent accuracy, ~2.6x
        # faster than "f32" = exact fp32 MFMA); "f16f8" (default) = f16x3 with the two correction products of the
        # MFMA-bound convs (32x32- and 16x16-pixel levels at full batch) on the fp8 matrix path, another 2-3 %
        # faster. All pass the 1e-3 parity bar against the reference: ~4e-6 (f32, f16x3) and ~2e-5 (f16f8) over the
        # 1000-step headline run (tests/test_gpu_round3.py, tests/test_gpu_f16f8.py).
        self.precision = os.environ.get("SR3_PRECISION", "f16f8")
        # split-f16 range policy: False (default) = a call whose activations leave the fp16 range is finished in
        # exact f32 with an `Sr3RangeWarning`; True = it raises `Sr3Error` (sr3_set_range_policy)
        self.strict_range = bool(int(os.environ.get("SR3_STRICT_RANGE", "0")))

    # ---- engine management -------------------------------------------------------------------
    def _device_index(self) -> int:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise Sr3Error("UNet parameters are on %s: the SR3 HIP path needs a GPU (call .cuda()); "
                           "there is no CPU fallback" % dev)
        return dev.index if dev.index is not None else torch.cuda.current_device()

    def engine(self) -> Engine:
        idx = self._device_index()
        if self._engine is None or self._engine.device != idx:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(self.cfg, idx)
            self._synced = {}
        self._sync_weights()
        if getattr(self._engine, "precision", None) != self.precision:
            self._engine.set_precision(self.precision)
        # Stream ordering, made explicit: on a non-default torch stream the library enqueues on that very
        # stream. On the default (null) stream it uses its own stream (hipGraph capture is not allowed
        # on the null stream), so pending torch work is waited for here and `finish()` waits for the
        # library before torch (or a collective) touches the results — both as device-side event waits
        # (`ready()` / `finish()`), not host synchronisations.
        cur = torch.cuda.current_stream(idx)
        self._own_stream = cur.cuda_stream == 0
        self._engine.set_stream(cur.cuda_stream)
        if getattr(self._engine, "_strict", None) != self.strict_range:
            self._engine.set_range_policy(self.strict_range)
            self._engine._strict = self.strict_range
        return self._engine

    def ready(self) -> None:
        """Call right before enqueueing library work that reads tensors torch produced (after the last
        `.to()` / `.contiguous()` of the inputs): the library's stream waits on the device for torch's stream —
        an event, no host stall."""
        if self._engine is not None and getattr(self, "_own_stream", True):
            self._engine.wait_for_stream(0)

    def finish(self) -> None:
        """Call after enqueueing library work whose results torch will read (see engine()): torch's stream
        waits on the device for the library's stream — results are ordered for every later torch operation
        (including `.cpu()` and collectives enqueued from torch's current stream) without a host synchronisation."""
        if self._engine is not None and getattr(self, "_own_stream", True):
            self._engine.stream_wait_for_engine(0)

    def _sync_weights(self) -> None:
        for name, p in self.named_parameters():
            sig = (p.data_ptr(), p._version)
            if self._synced.get(name) != sig:
                self._engine.load_weight(name, p.detach().to("cpu", torch.float32).contiguous().numpy())
                self._synced[name] = sig

    # ---- reference API -----------------------------------------------------------------------
    @torch.no_grad()
    def forward(self, x: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        eng = self.engine()
        if x.dim() != 4 or x.shape[1] != self.cfg.in_channel:
            raise RuntimeError(f"expected input [B, {self.cfg.in_channel}, H, W], got {tuple(x.shape)}")
        B, _, H, W = x.shape
        x = x.to(torch.float32).contiguous()
        nl = time.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
        if nl.numel() != B:
            raise RuntimeError(f"noise level must have {B} entries, got {nl.numel()}")
        out = torch.empty((B, self.cfg.out_channel, H, W), dtype=torch.float32, device=x.device)
        self.ready()
        eng.unet_forward(x.data_ptr(), nl.data_ptr(), B, H, W, out.data_ptr())
        self.finish()
        return out
