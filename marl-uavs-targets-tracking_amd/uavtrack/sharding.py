"""Multi-GPU layout: environments are independent (each reference Environment owns its
own lists, src/environment.py:36-43), so the batch is split by rank with no per-step
traffic.  The only exchange is the end-of-rollout gather of the per-environment episode
accumulators (train.py:181-192), one all-gather over RCCL/xGMI (gloo on CPU tensors)."""
from __future__ import annotations

from typing import Optional, Tuple


def shard_range(n_envs_total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """(first global env id, env count) of `rank`; the remainder goes to the low ranks."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, r = divmod(n_envs_total, world_size)
    count = q + (1 if rank < r else 0)
    offset = rank * q + min(rank, r)
    return offset, count


def gather_rollout_summary(ep_sums, n_envs_total: Optional[int] = None, group=None):
    """All-gather the [B_local, 5] episode accumulators into [B_total, 5], ordered by
    global env id.  Uneven shards (n_envs_total % world != 0) are padded for the
    collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return ep_sums
    world = dist.get_world_size(group)
    if world == 1:
        return ep_sums
    b_local = ep_sums.shape[0]
    if n_envs_total is None:
        n_envs_total = b_local * world
    b_max = -(-n_envs_total // world)
    send = ep_sums
    if b_local != b_max:
        send = torch.zeros((b_max,) + tuple(ep_sums.shape[1:]), dtype=ep_sums.dtype, device=ep_sums.device)
        send[:b_local] = ep_sums
    dev = ep_sums.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        send = send.cpu()          # rehearsal / debugging on gloo: stage through the host
    out = torch.empty((world * b_max,) + tuple(ep_sums.shape[1:]), dtype=ep_sums.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send.contiguous(), group=group)
    out = out.to(dev)
    if n_envs_total == world * b_max:
        return out
    parts = []
    for r in range(world):
        _, cnt = shard_range(n_envs_total, r, world)
        parts.append(out[r * b_max:r * b_max + cnt])
    return torch.cat(parts, dim=0)
