"""Data-parallel helpers: one process per GPU, pages/lines are independent.

The reference scales by whole-process replicas pinned round-robin to GPUs
(marie/orchestrate/deployments/__init__.py:1355-1411); here the same partition is explicit:
rank r of R takes items r, r+R, ... and the only collectives are a start-up broadcast of the packed
weight arena and a gather of the (small) decoded results.
"""
from __future__ import annotations

from typing import Any, List, Sequence


def shard_indices(n_items: int, rank: int, world: int) -> List[int]:
    """Static round-robin partition: rank r owns items r, r+world, ..."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} not in [0,{world})")
    return list(range(rank, n_items, world))


def gather_in_order(local: Sequence[Any], n_items: int, dist=None) -> List[Any]:
    """Gather per-rank result lists produced under ``shard_indices`` back into item order on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if len(local) != n_items:
            raise ValueError("single-rank gather: result count != item count")
        return list(local)
    world = dist.get_world_size()
    parts: List[Any] = [None] * world
    dist.all_gather_object(parts, list(local))
    out: List[Any] = [None] * n_items
    for r, part in enumerate(parts):
        idx = shard_indices(n_items, r, world)
        if len(part) != len(idx):
            raise RuntimeError(f"rank {r} returned {len(part)} results for {len(idx)} items")
        for i, v in zip(idx, part):
            out[i] = v
    return out


def broadcast_arena(model, ctx, dist, src: int = 0) -> None:
    """RCCL broadcast of the packed weight arena (rank ``src`` has called ``load_state``; the others
    ``alloc_arena``).  One buffer, one collective, start-up only."""
    import torch

    ptr, nbytes = model.arena()
    buf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    if dist.get_rank() == src:
        ctx.memcpy_dev(buf.data_ptr(), ptr, nbytes)
        ctx.synchronize()
    dist.broadcast(buf, src=src)
    if dist.get_rank() != src:
        ctx.memcpy_dev(ptr, buf.data_ptr(), nbytes)
    ctx.synchronize()
    torch.cuda.synchronize()


def broadcast_arenas(model, ctx, dist, src: int = 0, device: str = "cuda") -> None:
    """Same as :func:`broadcast_arena` for models that keep several arenas (``model.arenas()`` -> [(ptr, bytes), ...]:
    the DiT detector and TrOCR keep the ViT encoder and the heads / decoder apart).  ``device="cpu"`` is the gloo
    rehearsal path of the tests (``ctx.memcpy_dev`` then copies between host buffers)."""
    import torch

    for ptr, nbytes in model.arenas():
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if dist.get_rank() == src:
            ctx.memcpy_dev(buf.data_ptr(), ptr, nbytes)
            ctx.synchronize()
        dist.broadcast(buf, src=src)
        if dist.get_rank() != src:
            ctx.memcpy_dev(ptr, buf.data_ptr(), nbytes)
        ctx.synchronize()
    if device == "cuda":
        torch.cuda.synchronize()


def arenas_checksum(model, ctx, device: str = "cuda") -> int:
    """Sum of all arena bytes (int64) — lets every rank confirm it holds rank 0's weights after the broadcast."""
    import torch

    total = 0
    for ptr, nbytes in model.arenas():
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        ctx.memcpy_dev(buf.data_ptr(), ptr, nbytes)
        ctx.synchronize()
        total += int(buf.to(torch.int64).sum().item())
    return total
