"""Checkpoint ingest for the SR3 sampler.

Two on-disk layouts exist in the reference:
  * upstream SR3 generator files `<prefix>_gen.pth`: a plain state_dict of the GaussianDiffusion
    module (`denoise_fn.*` weights + the 12 schedule buffers), written by model/sr/model.py:139-162
    and read by model/sr/model.py:164-195 and lib/trainer_temp.py:196-209;
  * the fork's combined training checkpoint (lib/trainer_temp.py:243-262): a dict whose
    'sr_model_state' entry is that same state_dict (next to MICA / optimizer / scheduler states),
    read by lib/trainer_temp.py:170-178.
Both are loaded with `torch.load(..., weights_only=True)` (nothing in the file is executed), a
DataParallel / DDP `module.` prefix is dropped (lib/trainer_temp.py:175-176 adds it for wrapped
models; the HIP sampler is never wrapped), and the result goes through `load_state_dict` with the
reference's `strict=False` semantics: unknown and missing keys are reported, shape mismatches raise.
"""
from __future__ import annotations

import os
from typing import Dict, List, NamedTuple, Optional

import torch
from torch import nn

SR_STATE_KEY = "sr_model_state"          # lib/trainer_temp.py:246


class IngestReport(NamedTuple):
    path: str
    layout: str                  # "gen" | "combined"
    loaded: int                  # tensors copied into the model
    missing_keys: List[str]
    unexpected_keys: List[str]
    epoch: Optional[int]
    global_step: Optional[int]


def resolve_path(path_or_prefix: str) -> str:
    """`pretrained_model_path` in the yml files is a prefix ('.../I640000_E37'): the reference
    appends '_gen.pth' (lib/trainer_temp.py:199). A full file name is accepted too."""
    if os.path.isfile(path_or_prefix):
        return path_or_prefix
    gen = "{}_gen.pth".format(path_or_prefix)
    if os.path.isfile(gen):
        return gen
    raise FileNotFoundError(f"no checkpoint at '{path_or_prefix}' or '{gen}'")


def _strip_module(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}


def read_sr_state_dict(path_or_prefix: str):
    """-> (state_dict, layout, epoch, global_step); tensors on CPU."""
    path = resolve_path(path_or_prefix)
    obj = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(obj, dict):
        raise ValueError(f"{path}: expected a dict, got {type(obj).__name__}")
    if SR_STATE_KEY in obj:
        sd, layout = obj[SR_STATE_KEY], "combined"
        epoch, step = obj.get("epoch"), obj.get("global_step")
    else:
        sd, layout, epoch, step = obj, "gen", None, None
    bad = [k for k, v in sd.items() if not isinstance(v, torch.Tensor)]
    if bad:
        raise ValueError(f"{path}: non-tensor entries in the SR state dict: {bad[:4]}")
    return _strip_module(sd), layout, epoch, step, path


def load_sr_checkpoint(model: nn.Module, path_or_prefix: str, strict: bool = False) -> IngestReport:
    """Load an SR3 checkpoint into `model` (a GaussianDiffusion from define_G, or a bare UNet — for a
    UNet the `denoise_fn.` prefix is dropped and schedule buffers are ignored)."""
    sd, layout, epoch, step, path = read_sr_state_dict(path_or_prefix)
    if not hasattr(model, "denoise_fn"):
        sd = {k[len("denoise_fn."):]: v for k, v in sd.items() if k.startswith("denoise_fn.")}
    own = model.state_dict()
    res = model.load_state_dict(sd, strict=strict)
    loaded = sum(1 for k in sd if k in own)
    return IngestReport(path, layout, loaded, list(res.missing_keys), list(res.unexpected_keys),
                        epoch, step)


def save_gen(model: nn.Module, prefix: str) -> str:
    """Write `<prefix>_gen.pth` in the upstream layout (model/sr/model.py:139-153): CPU tensors,
    plain state_dict. Used for round-trip tests and for exporting synthetic weights."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    path = "{}_gen.pth".format(prefix)
    torch.save(sd, path)
    return path
