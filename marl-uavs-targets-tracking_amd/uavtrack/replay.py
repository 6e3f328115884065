"""Device-resident replay buffers: the reference's `ReplayBuffer` / `PrioritizedReplayBuffer`
(src/train.py:41-139) hold Python tuples in a deque and are filled one UAV transition at a time
(train.py:176-180, 194).  Here a buffer is four preallocated device tensors used as a ring, filled with the
[T, B, N] outputs of a rollout in one indexed copy, and sampled with one gather -- nothing leaves the GPU.

Semantics kept from the reference: capacity-bounded FIFO overwrite (deque(maxlen) / `pos` ring); uniform
sampling WITHOUT replacement of min(batch, size) transitions (random.sample, train.py:57); prioritised sampling
WITH replacement from p_i^alpha / sum (np.random.choice, train.py:106), new transitions enter at the current
maximum priority (train.py:87-96), importance weights (size * P(i))^-beta / max (train.py:109-112).
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

KEYS = ("states", "actions", "rewards", "next_states")


def transitions_from_rollout(obs_in: torch.Tensor, out: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The (state, action, reward, next_state) tuples of train.py:176-180 from one rollout's outputs:
    obs_in [B,N,12] is what the policy saw first; out = {obs [T,B,N,12], actions [T,B,N], reward [T,B,N]}."""
    obs = out["obs"]
    states = torch.cat([obs_in.unsqueeze(0), obs[:-1]], dim=0)
    D = obs.shape[-1]
    return {"states": states.reshape(-1, D), "actions": out["actions"].reshape(-1),
            "rewards": out["reward"].reshape(-1), "next_states": obs.reshape(-1, D)}


class DeviceReplayBuffer:
    """ReplayBuffer (train.py:41-70) as a device ring."""

    def __init__(self, capacity: int, device, obs_dim: int = 12):
        self.capacity = int(capacity)
        self.device = torch.device(device)
        self.store = {"states": torch.empty(self.capacity, obs_dim, device=self.device),
                      "actions": torch.empty(self.capacity, dtype=torch.int32, device=self.device),
                      "rewards": torch.empty(self.capacity, device=self.device),
                      "next_states": torch.empty(self.capacity, obs_dim, device=self.device)}
        self.pos = 0          # next slot to write
        self.count = 0        # valid transitions

    def size(self) -> int:
        return self.count

    def _slots(self, n: int) -> Tuple[torch.Tensor, int]:
        """Ring slots for n new transitions (only the last `capacity` of them survive, like deque(maxlen))."""
        skip = max(0, n - self.capacity)
        start = (self.pos + skip) % self.capacity
        idx = (torch.arange(n - skip, device=self.device) + start) % self.capacity
        return idx, skip

    def add(self, transition_dict: Dict[str, torch.Tensor]) -> torch.Tensor:
        """transition_dict: states [n,12], actions [n], rewards [n], next_states [n,12] (any leading shape is
        flattened).  Returns the ring slots written."""
        D = self.store["states"].shape[1]
        n = transition_dict["actions"].numel()
        idx, skip = self._slots(n)
        for k in KEYS:
            src = transition_dict[k].to(self.device)
            src = src.reshape(n, D) if k.endswith("states") else src.reshape(n)
            self.store[k].index_copy_(0, idx, src[skip:].to(self.store[k].dtype))
        self.pos = (self.pos + n) % self.capacity
        self.count = min(self.capacity, self.count + n)
        return idx

    def sample(self, batch_size: int, generator: Optional[torch.Generator] = None) -> Dict[str, torch.Tensor]:
        k = min(int(batch_size), self.count)
        idx = torch.randperm(self.count, device=self.device, generator=generator)[:k]
        return {key: self.store[key][idx] for key in KEYS}


class PrioritizedDeviceReplayBuffer(DeviceReplayBuffer):
    """PrioritizedReplayBuffer (train.py:73-139) as a device ring."""

    def __init__(self, capacity: int, device, alpha: float = 0.6, obs_dim: int = 12):
        super().__init__(capacity, device, obs_dim)
        self.alpha = float(alpha)
        self.priorities = torch.zeros(self.capacity, device=self.device)

    def add(self, transition_dict: Dict[str, torch.Tensor]) -> torch.Tensor:
        # every transition of the call enters at the maximum priority seen so far (1.0 for an empty buffer);
        # within one reference add() the maximum cannot grow, so one value serves the whole batch
        top = self.priorities.max() if self.count > 0 else torch.ones((), device=self.device)
        idx = super().add(transition_dict)
        self.priorities.index_fill_(0, idx, 0.0)
        self.priorities.index_add_(0, idx, top.expand(idx.numel()))
        return idx

    def sample(self, batch_size: int, beta: float = 0.4, generator: Optional[torch.Generator] = None):
        if self.count == 0:
            return {k: self.store[k][:0] for k in KEYS}, None, None
        prob = self.priorities[:self.count] ** self.alpha
        prob = prob / prob.sum()
        k = min(int(batch_size), self.count)
        idx = torch.multinomial(prob, k, replacement=True, generator=generator)
        weights = (self.count * prob[idx]) ** (-beta)
        weights = weights / weights.max()
        return {key: self.store[key][idx] for key in KEYS}, idx, weights

    def update_priorities(self, batch_indices: torch.Tensor, batch_priorities: torch.Tensor) -> None:
        self.priorities.index_copy_(0, batch_indices.to(self.device), batch_priorities.to(self.device, torch.float32))
