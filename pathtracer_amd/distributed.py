"""Multi-GPU split: one process per GPU, each renders a disjoint subset of the samples of every
pixel (or, `subset_tiles`, every sample of a disjoint subset of the 8x8-pixel tiles); the partial per-pixel sums are combined with one reduce(sum) to rank 0 (RCCL over xGMI when the
tensors are on GPUs; the same code runs over gloo on CPU tensors in the tests).

Rank g of G renders the global sample indices g, g + G, g + 2G, ...  The sub-pixel table and the
sampler keys are indexed by the GLOBAL sample index, so the union over ranks is the sample set of a
single-GPU render whatever G is; only the order of floating-point additions differs.
"""
from __future__ import annotations

from dataclasses import replace
from typing import Callable, Optional

import torch
import torch.distributed as dist

from .renderer import RenderSettings


def subset_settings(settings: RenderSettings, rank: int, world: int) -> RenderSettings:
    """Settings for rank `rank` of `world`: same frame, samples rank, rank + world, ..."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return replace(settings, sampleBegin=rank, sampleStep=world)


def subset_tiles(settings: RenderSettings, rank: int, world: int) -> RenderSettings:
    """The alternative split for high resolutions / few samples: rank `rank` renders every sample of the 8x8-pixel
    tiles rank, rank + world, ... (interleaved, so that every rank sees every part of the picture) and leaves the other
    pixels at zero — the same reduce(sum) combines the ranks, each pixel receiving exactly one non-zero term."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    return replace(settings, tileBegin=rank, tileStep=world)


def samples_of_rank(total_spp: int, rank: int, world: int) -> int:
    return len(range(rank, total_spp, world))


def combine_partial_sums(buf: torch.Tensor, total_spp: int, rank: int, world: int,
                         resolve: Optional[Callable[[torch.Tensor], None]] = None) -> Optional[torch.Tensor]:
    """Reduce the (H, W, 4) partial-sum buffers of all ranks onto rank 0 and turn the sum into the mean
    image there (rgb / total_spp, alpha 1).  Returns the image on rank 0, None elsewhere."""
    if world > 1:
        if buf.is_cuda and dist.get_backend() == "gloo":  # gloo reduces host tensors only (rehearsals without RCCL)
            host = buf.cpu()
            dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
            buf.copy_(host)
        else:
            dist.reduce(buf, dst=0, op=dist.ReduceOp.SUM)
    if rank != 0:
        return None
    if resolve is not None:
        resolve(buf)
    else:
        buf[..., :3] /= float(total_spp)
        buf[..., 3] = 1.0
    return buf
