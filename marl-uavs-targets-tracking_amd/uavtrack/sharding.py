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


class SummaryGather:
    """Handle of one in-flight end-of-rollout gather (gather_rollout_summary_async).  `wait()` returns the
    [B_total, 5] tensor ordered by global env id; on RCCL it makes the *current stream* wait, not the host."""

    def __init__(self, work, out, dev, n_envs_total, world, b_max, passthrough=None):
        self._work, self._out, self._dev = work, out, dev
        self._n, self._world, self._b_max = n_envs_total, world, b_max
        self._result = passthrough

    def wait(self):
        if self._result is not None:
            return self._result
        if self._work is not None:
            self._work.wait()
        out = self._out.to(self._dev)
        if self._n != self._world * self._b_max:
            parts = []
            for r in range(self._world):
                _, cnt = shard_range(self._n, r, self._world)
                parts.append(out[r * self._b_max:r * self._b_max + cnt])
            out = torch_cat(parts)
        self._result = out
        return out


def torch_cat(parts):
    import torch
    return torch.cat(parts, dim=0)


def gather_rollout_summary_async(ep_sums, n_envs_total: Optional[int] = None, group=None) -> SummaryGather:
    """Start the all-gather of the [B_local, 5] episode accumulators and return at once: the collective runs on
    RCCL's own stream from a private copy of `ep_sums`, so the next rollout (which overwrites `ep_sums`) can be
    launched immediately and overlaps the exchange and any skew between ranks."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return SummaryGather(None, None, None, 0, 1, 0, passthrough=ep_sums)
    world = dist.get_world_size(group)
    b_local = ep_sums.shape[0]
    if n_envs_total is None:
        n_envs_total = b_local * world
    b_max = -(-n_envs_total // world)
    send = torch.zeros((b_max,) + tuple(ep_sums.shape[1:]), dtype=ep_sums.dtype, device=ep_sums.device)
    send[:b_local] = ep_sums           # private copy (also the padding of an uneven shard)
    dev = ep_sums.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        send = send.cpu()          # rehearsal / debugging on gloo: stage through the host
    out = torch.empty((world * b_max,) + tuple(ep_sums.shape[1:]), dtype=ep_sums.dtype, device=send.device)
    work = dist.all_gather_into_tensor(out, send.contiguous(), group=group, async_op=True)
    return SummaryGather(work, out, dev, n_envs_total, world, b_max)


def gather_rollout_summary(ep_sums, n_envs_total: Optional[int] = None, group=None):
    """All-gather the [B_local, 5] episode accumulators into [B_total, 5], ordered by
    global env id.  Uneven shards (n_envs_total % world != 0) are padded for the
    collective and trimmed afterwards."""
    return gather_rollout_summary_async(ep_sums, n_envs_total, group).wait()
