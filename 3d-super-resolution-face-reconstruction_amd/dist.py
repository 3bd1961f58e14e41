"""Multi-GPU: one process per GPU, the image batch sharded in contiguous slices, no communication
inside the T-step loop, ONE all-gather of the finished images at the end (RCCL over xGMI when the
backend is "nccl"; "gloo" in the CPU tests). The reference itself is single-process
(SURVEY.md §0, §8e); every image's chain is independent, so this is the whole exchange.

RNG: rank r samples images [start, stop) with `image_offset=start`, so the device Philox stream of
image i is the same whatever the world size.
"""
from __future__ import annotations

import os
from typing import Callable, List, Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def force_collective() -> bool:
    """SR3_FORCE_COLLECTIVE=1: take the collective path even with ONE rank (the RCCL all-gather then runs as a
    world-size-1 collective). Exists so that a single-GPU box exercises librccl initialisation and the product's
    collective call before the first multi-GPU run."""
    return os.environ.get("SR3_FORCE_COLLECTIVE", "0") not in ("", "0")


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """Initialises torch.distributed from RANK/WORLD_SIZE/MASTER_* (torch.distributed.run)."""
    rank, world, local = env_rank()
    if (world > 1 or force_collective()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice of n images for `rank`; the first n % world ranks get one extra."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_gather_images(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Gathers the per-rank [b_r, ...] slices (contiguous, shard_bounds order) into [n_total, ...]
    on every rank with a single collective."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective()):
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    counts = [shard_bounds(n_total, world, r) for r in range(world)]
    sizes = [b - a for a, b in counts]
    tail = tuple(local.shape[1:])
    if len(set(sizes)) == 1:
        out = local.new_empty((n_total,) + tail)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # ragged tail: pad every slice to the largest, still one collective
    m = max(sizes)
    padded = local.new_zeros((m,) + tail)
    padded[: local.shape[0]] = local
    out = local.new_empty((world * m,) + tail)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(world)], dim=0)


def sharded_super_resolution(sample_fn: Callable[[torch.Tensor, int], torch.Tensor],
                             x_full: torch.Tensor, gather: bool = True) -> torch.Tensor:
    """Runs `sample_fn(x_local, image_offset)` on this rank's slice of `x_full` [N,3,H,W] and
    all-gathers the results. `sample_fn` is `lambda x, off: netG.super_resolution_batch(x,
    seed=seed, image_offset=off)` in production."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    a, b = shard_bounds(x_full.shape[0], world, rank)
    local = sample_fn(x_full[a:b], a)
    if local.is_cuda:
        # explicit dependency before the collective: the sampler may have run on the library's own
        # stream, which RCCL's stream does not wait for (the torch facade already waits for it; a raw
        # Engine-based sample_fn may not): wait for every stream of this device, once per call
        torch.cuda.synchronize(local.device)
    return all_gather_images(local, x_full.shape[0]) if gather else local
