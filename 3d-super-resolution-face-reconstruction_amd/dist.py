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


def _local_shard(x_full: torch.Tensor):
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    return shard_bounds(x_full.shape[0], world, rank)


def sharded_super_resolution(sample_fn: Callable[[torch.Tensor, int], torch.Tensor],
                             x_full: torch.Tensor, gather: bool = True) -> torch.Tensor:
    """Runs `sample_fn(x_local, image_offset)` on this rank's slice of `x_full` [N,3,H,W] and
    all-gathers the results. `sample_fn` is `lambda x, off: netG.super_resolution_batch(x,
    seed=seed, image_offset=off)` in production.

    Stream ordering: the result must be ordered on torch's CURRENT stream of its device when `sample_fn` returns — the
    torch facade guarantees it (UNet.finish(): torch's stream waits on the device for the library's stream, an event,
    no host stall), and the collective is enqueued behind that stream like any torch operation. A `sample_fn` built on
    a raw `Engine` calls `eng.stream_wait_for_engine(torch.cuda.current_stream().cuda_stream)` itself. No host
    synchronisation happens here."""
    a, b = _local_shard(x_full)
    local = sample_fn(x_full[a:b], a)
    return all_gather_images(local, x_full.shape[0]) if gather else local


def sharded_p_sample_loop(netG, x_full: torch.Tensor, continous: bool = False, seed: int = 0) -> torch.Tensor:
    """The reference's `super_resolution(x_in, continous)` (diffusion.py:189-215, 223-225) for a batch sharded over the
    ranks, with the reference's return convention on EVERY rank:

      continous=False -> `ret_img[-1]`: the LAST image of the global batch, [3,H,W]
      continous=True  -> `ret_img` = cat([x_in, frame_0, ..., frame_9]) on dim 0, [(1 + n_frames) * N, 3, H, W], the
                         conditioning batch first (:203-204), then the whole batch after every recorded step (:209-211)

    Two collectives at most: the final images (N/world x 3 x r x r per rank) and, if continous, the frames (n_frames
    times as much: 126 MB per rank for 256 images at 128x128 — still one call). `seed` must be the same on every rank
    (Philox streams are keyed by the global image index)."""
    N = x_full.shape[0]
    a, b = _local_shard(x_full)
    x_loc = x_full[a:b].to(next(netG.parameters()).device)
    if not continous:
        out = all_gather_images(netG.super_resolution_batch(x_loc, seed=seed, image_offset=a), N)
        return out[-1]
    out, frames = netG.sample_batch(x_loc, True, None, seed, a)            # frames [n, b_local, C, H, W]
    # gather along the image axis: put it first, one collective, back
    fr = all_gather_images(frames.transpose(0, 1).contiguous(), N)       # [N, n, C, H, W]
    fr = fr.transpose(0, 1).reshape(-1, *frames.shape[2:])               # [n * N, C, H, W], frame-major like ret_img
    return torch.cat([x_full.to(device=fr.device, dtype=torch.float32), fr], dim=0)
