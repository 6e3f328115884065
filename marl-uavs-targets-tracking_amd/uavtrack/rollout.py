"""Closed-loop batched rollout driver: the caller side of the hot path.

The reference's `train.operate_epoch` (src/train.py:142-196) runs, per step, N batch-1 actor
forwards with a host round trip each (`actor_critic.py:138-148`), then `env.step`.  Here one
`[B*N, 12]` actor forward, a categorical sample and `uavtrack_step` are chained on one stream
with nothing returning to the host, and `steps_per_graph` such steps are captured once into a
HIP graph (`torch.cuda.CUDAGraph`; the library's launches are stream-ordered and allocation-free,
so they capture like any other kernel) and replayed -- a per-step launch sequence is otherwise
host-bound at this batch size.  Episode accumulators follow train.py:181-192.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

from .env import BatchedUavEnv


class ActorMLP(torch.nn.Module):
    """Same shape as the reference's shared actor `FnnPolicyNet` (src/models/actor_critic.py:85-98):
    Linear(12, hidden) - ReLU - Linear(hidden, na) - softmax.  Weights load from its state_dict."""

    def __init__(self, state_dim: int = 12, hidden_dim: int = 256, action_dim: int = 12):
        super().__init__()
        self.fc1 = torch.nn.Linear(state_dim, hidden_dim)
        self.fc2 = torch.nn.Linear(hidden_dim, action_dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return torch.softmax(self.fc2(torch.relu(self.fc1(x))), dim=-1)


def sample_actions(probs: torch.Tensor) -> torch.Tensor:
    """Categorical(probs).sample() for [B, N, na] -> int32 [B, N] (actor_critic.py:145-147), drawn with
    the exponential-race form argmax_i p_i / E_i, E_i ~ Exp(1): three elementwise kernels instead of
    torch.multinomial's per-row search."""
    return (probs / torch.empty_like(probs).exponential_()).argmax(dim=-1).to(torch.int32)


class BatchedRollout:
    """policy = callable obs[B,N,12] -> probs[B,N,na] (e.g. ActorMLP), or the string "greedy" for the
    reference's C-METHOD baseline (uav.py:324-369) computed by the library's own policy kernel.

    device_actor=True (policy must then have FnnPolicyNet's parameters, e.g. ActorMLP or the reference's own
    actor): the policy runs inside the library -- per step as uavtrack_actor_actions, or, with `run_fused`,
    actor and environment together in ONE launch for the whole rollout (uavtrack_run_actor); call
    `sync_actor()` after every learner update to upload the new weights."""

    def __init__(self, env: BatchedUavEnv, policy, select: Callable[[torch.Tensor], torch.Tensor] = sample_actions,
                 steps_per_graph: int = 8, use_graph: bool = True, seed: int = 0, device_actor: bool = False,
                 fuse_chunks: bool = False):
        """fuse_chunks (library policies only: device_actor or "greedy"): run() issues every `steps_per_graph` steps as
        ONE launch with the policy inside the step kernel (uavtrack_run_actor / uavtrack_run_greedy with T = k -- the
        same bits as k x (policy kernel, step), see tests) instead of replaying a graph of 2 k kernels: a kernel boundary
        costs this path more than the step itself (a T = 1 launch is ~7 us of GPU time for a 2.6 us step)."""
        self.env, self.policy, self.select, self.seed = env, policy, select, seed
        self.device_actor = device_actor
        self.fuse_chunks = bool(fuse_chunks) and (device_actor or isinstance(policy, str))
        self._chunk_out: Optional[Dict[str, torch.Tensor]] = None
        if device_actor:
            env.set_actor(policy)
        self.k = max(1, int(steps_per_graph))
        self.use_graph = use_graph
        B = env.B
        dev = env.device
        self.obs = torch.empty(B, env.N, 12, device=dev)               # static buffers (graph I/O)
        self.ep = torch.zeros(B, 5, device=dev)                        # sum_t mean_i reward, 3 terms, covered
        self.last_reward = torch.empty(B, env.N, device=dev)
        self._graph: Optional[torch.cuda.CUDAGraph] = None
        self._ep_ring: Optional[torch.Tensor] = None                   # fuse_chunks: one row of episode sums per chunk of a run()
        self._ep_used = 0
        self._bound: Dict[int, Callable] = {}
        self._chunk_ready = False

    # one closed-loop step on the current stream; everything stays on the device
    def _one_step(self):
        if isinstance(self.policy, str):
            assert self.policy == "greedy", self.policy
            actions = self.env.greedy_actions(self.seed)     # draws are keyed by (seed, env, step_count, uav)
        elif self.device_actor:
            actions = self.env.actor_actions(self.obs, self.seed)
        else:
            with torch.no_grad():
                actions = self.select(self.policy(self.obs))
        # the kernel writes the static buffers in place and adds to the accumulators: no torch op per step.
        # (The policy has consumed self.obs before the step kernel overwrites it: same stream.)
        self.env.step(actions, ep_sums=self.ep, out_obs=self.obs, out_reward=self.last_reward)

    def _capture(self):
        # warm up on a side stream (allocator pools, lazy inits), then capture k steps
        s = torch.cuda.Stream(device=self.env.device)
        s.wait_stream(torch.cuda.current_stream(self.env.device))
        with torch.cuda.stream(s):
            state = self.env.get_state()
            obs0, ep0 = self.obs.clone(), self.ep.clone()
            for _ in range(2):
                self._one_step()
            self.env.set_state(**state)                    # undo the warm-up steps
            self.obs.copy_(obs0); self.ep.copy_(ep0)
        torch.cuda.current_stream(self.env.device).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        state = self.env.get_state()
        obs0, ep0 = self.obs.clone(), self.ep.clone()
        with torch.cuda.graph(g):
            for _ in range(self.k):
                self._one_step()
        self.env.set_state(**state)                        # capture does not execute; be explicit anyway
        self.obs.copy_(obs0); self.ep.copy_(ep0)
        self._graph = g

    def reset(self, seed: int = 0):
        self.obs.copy_(self.env.reset(seed=seed))
        self.ep.zero_()

    def run(self, steps: int) -> Dict[str, torch.Tensor]:
        """Advance `steps` closed-loop steps; returns the episode accumulators so far."""
        done = 0
        if self.fuse_chunks:
            self._ep_used = 0
            while steps - done >= self.k:
                self._fused_chunk(self.k)
                done += self.k
            if steps - done:
                self._fused_chunk(steps - done)
            if self._ep_used:          # the chunks' episode sums, added up ONCE per run() (one small kernel, not one per chunk)
                self.ep += self._ep_ring[:self._ep_used].sum(dim=0)
                self._ep_used = 0
            return {"ep_sums": self.ep, "obs": self.obs, "reward": self.last_reward}
        if self.use_graph and steps >= self.k:
            if self._graph is None:
                self._capture()
            while steps - done >= self.k:
                self._graph.replay()
                done += self.k
        for _ in range(steps - done):
            self._one_step()
        return {"ep_sums": self.ep, "obs": self.obs, "reward": self.last_reward}

    def _fused_chunk(self, k: int):
        """k closed-loop steps of the library policy + environment in one launch; state of the driver as after k _one_step().
        Nothing but the launch is issued per chunk: `obs` / `last_reward` become VIEWS of the chunk's last rows (the next
        launch reads its first observation from there -- each lane reads its own row once, before it writes anything, and
        that row is only rewritten by the same lane at the chunk's last step), and the chunk's episode sums go to their own
        row of a ring that run() adds up at the end.  (Three small torch kernels per chunk cost as much as two steps.)"""
        ring = self._ep_ring
        if ring is None or self._ep_used >= ring.shape[0]:
            if ring is not None:                      # a very long run(): fold what the ring holds and start over
                self.ep += ring[:self._ep_used].sum(dim=0)
            else:
                self._ep_ring = ring = torch.empty(256, self.env.B, 5, device=self.env.device)
            self._ep_used = 0
        ep_row = ring[self._ep_used]
        self._ep_used += 1
        out = self._chunk_out if self._chunk_out is not None and self._chunk_out["reward"].shape[0] == k else None
        if out is not None and k == self.k and self._chunk_ready:
            # steady state: a pre-built library call per ring row (the chunk's buffers, its first observation = the previous
            # chunk's last row, its episode-sum row are all fixed addresses)
            call = self._bound.get(self._ep_used - 1)
            if call is None:
                call = self._bound[self._ep_used - 1] = self.env.bind_run(
                    k, dict(out, ep_sums=ep_row), "actor" if self.device_actor else "greedy", obs_in=self.obs, seed=self.seed)
            call()
            return
        if out is not None:
            out = dict(out, ep_sums=ep_row)
        if self.device_actor:
            res = self.env.run_actor(k, self.obs, seed=self.seed, want_terms=True, out=out)
        else:
            res = self.env.run_greedy(k, seed=self.seed, want_actions=False, out=out)
        if out is None:
            ep_row.copy_(res["ep_sums"])
            if k == self.k:
                self._chunk_out = res
        self.obs = res["obs"][-1]
        self.last_reward = res["reward"][-1]
        self._chunk_ready = out is not None        # from now on obs / last_reward are the views the bound calls were built on

    def sync_actor(self):
        """Upload the policy's current parameters to the library (device_actor mode; after a learner update)."""
        self.env.set_actor(self.policy)

    def run_fused(self, steps: int, want_terms: bool = False, out: Optional[Dict[str, torch.Tensor]] = None
                  ) -> Dict[str, torch.Tensor]:
        """The same `steps` closed-loop steps as run(), as ONE launch (device_actor mode, MAAC / MAAC-G).  Returns
        the launch's outputs -- obs/actions/reward [T, ...] are the transitions of train.py:176-180 (state[t] is
        obs[t-1], or the observation held before the call for t = 0) -- plus the running episode sums."""
        assert self.device_actor, "run_fused needs device_actor=True"
        res = self.env.run_actor(steps, self.obs, seed=self.seed, want_terms=want_terms, out=out)
        self.obs.copy_(res["obs"][-1])
        self.last_reward.copy_(res["reward"][-1])
        self.ep += res["ep_sums"]
        res = dict(res)
        res["ep_sums"] = self.ep
        return res
