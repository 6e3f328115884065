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


# ---------------------------------------------------------------------------------------------------------------------
# The learner's side of a sharded rollout.  The reference has ONE learner that consumes (state, action, reward,
# next_state) transitions (train.py:176-194, 250-252); with the rollouts on R ranks it needs a sample of every rank's
# transitions, not the trajectories themselves (1 GB per rank and rollout at 4096 x 20 x 200).  Each rank draws K of its
# own transitions, packs them into one [K, 28] int32 block (bit views: nothing is converted) and ONE all-gather per rollout
# hands every rank -- the learner among them -- the R K sampled rows, ready for DeviceReplayBuffer.add().

TRANSITION_WORDS = 12 + 1 + 1 + 12 + 2      # state, action, reward, next_state, global transition id (int64)


def sample_local_transitions(obs_in, out, k: int, env_offset: int = 0, n_envs_total: Optional[int] = None, generator=None):
    """K transitions of this rank's rollout, uniformly without replacement: out = {obs [T,B,N,12], actions [T,B,N],
    reward [T,B,N]} (a step_many / run_actor result plus the actions that drove it), obs_in [B,N,12] what the policy saw
    first.  Returns {"states" [K,12], "actions" [K] int32, "rewards" [K], "next_states" [K,12], "index" [K] int64}; index =
    (t * B_total + env_offset + b) * N + i, the transition's place in the UNSHARDED rollout's [T, B_total, N] grid (for
    priorities, de-duplication, and the tests).  The [T+1] state array of transitions_from_rollout is never built."""
    import torch
    obs, act, rew = out["obs"], out["actions"], out["reward"]
    T, B, N = rew.shape
    n = T * B * N
    if not (0 < k <= n):
        raise ValueError(f"k = {k} outside (0, {n}] transitions of this rollout")
    if n_envs_total is None:
        n_envs_total = B
    dev = rew.device
    if 16 * k <= n:
        # k << n (4096 of 16.4 M at the reference shape): a permutation of all n would cost more than the rollout.  The
        # distinct values among m uniform draws are a uniform random subset of their size; a uniform k-subset of that set
        # is a uniform k-subset of the rollout, and the permutation that picks it also randomises the order.
        cand = torch.unique(torch.randint(0, n, (k + k // 4 + 64,), device=dev, generator=generator))
        pick = (cand[torch.randperm(cand.numel(), device=dev, generator=generator)[:k]] if cand.numel() >= k
                else torch.randperm(n, device=dev, generator=generator)[:k])
    else:
        pick = torch.randperm(n, device=dev, generator=generator)[:k]
    t, rem = pick // (B * N), pick % (B * N)
    b, i = rem // N, rem % N
    prev = obs[(t - 1).clamp(min=0), b, i]
    states = torch.where((t == 0).unsqueeze(1), obs_in[b, i], prev)
    return {"states": states, "actions": act[t, b, i].to(torch.int32), "rewards": rew[t, b, i],
            "next_states": obs[t, b, i], "index": (t * n_envs_total + env_offset + b) * N + i}


def _pack_transitions(s):
    import torch
    k = s["actions"].shape[0]
    return torch.cat([s["states"].contiguous().view(torch.int32), s["actions"].view(k, 1),
                      s["rewards"].contiguous().view(torch.int32).view(k, 1), s["next_states"].contiguous().view(torch.int32),
                      s["index"].contiguous().view(torch.int32).view(k, 2)], dim=1).contiguous()


def _unpack_transitions(blk):
    import torch
    return {"states": blk[:, 0:12].contiguous().view(torch.float32), "actions": blk[:, 12].contiguous(),
            "rewards": blk[:, 13].contiguous().view(torch.float32), "next_states": blk[:, 14:26].contiguous().view(torch.float32),
            "index": blk[:, 26:28].contiguous().view(torch.int64).view(-1)}


class TransitionGather:
    """Handle of one in-flight transition gather; wait() -> the dict of sample_local_transitions with R K rows, rank-major."""

    def __init__(self, work, out, dev, world=1, passthrough=None):
        self._work, self._out, self._dev, self._world, self._result = work, out, dev, world, passthrough

    @property
    def nbytes_per_rank(self) -> int:
        """bytes each rank contributes to the collective (0 without a process group)"""
        return 0 if self._out is None else self._out.numel() * 4 // self._world

    def wait(self):
        if self._result is None:
            if self._work is not None:
                self._work.wait()
            self._result = _unpack_transitions(self._out.to(self._dev))
        return self._result


def gather_transitions_async(sample, group=None) -> TransitionGather:
    """Start the all-gather of every rank's K sampled transitions (equal K on all ranks) and return at once; like
    gather_rollout_summary_async it runs from a private packed copy on RCCL's own stream."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return TransitionGather(None, None, None, passthrough=dict(sample))
    world = dist.get_world_size(group)
    send = _pack_transitions(sample)
    dev = send.device
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        send = send.cpu()          # rehearsal on gloo: stage through the host
    out = torch.empty((world * send.shape[0], TRANSITION_WORDS), dtype=torch.int32, device=send.device)
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=True)
    return TransitionGather(work, out, dev, world)


def gather_transitions(sample, group=None):
    """Blocking form: {"states" [R K, 12], "actions", "rewards", "next_states", "index"} on every rank."""
    return gather_transitions_async(sample, group).wait()
