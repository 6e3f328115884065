"""Drop-in stand-in for the reference's `Environment` class (src/environment.py:12), so
that train.operate_epoch-style loops and the MAAC / MAAC-R actors run unchanged:

    env = Environment(n_uav=, m_targets=, x_max=, y_max=, na=)      # main.py:55-59
    env.reset(config=config)                                         # train.py:232,311,383
    for uav in env.uav_list: state = uav.get_local_state()           # train.py:165-166
    next_states, reward, covered = env.step(config, pmi, actions)    # train.py:176
    env.n_uav, env.get_states(), env.position, env.covered_target_num,
    env.save_position(dir, i), env.save_covered_num(dir, i)          # train.py:187,287-288
    uav.get_action_by_direction(env.target_list, env.uav_list)       # train.py:350 (C-METHOD, train.run_epoch)

One reference environment is a batch of one on the GPU; every number `step` returns comes
from the HIP kernels through the C ABI (uavtrack.BatchedUavEnv).  Batched training should use
BatchedUavEnv directly.

Identical seeds -- for `reset` and for `step` with actions the CALLER supplies: the reference draws its reset
from Python's global `random` and every target burns one unused draw per step (target.py:34).  A batch of one
has no use for the device's Philox reset, so this adapter draws the reset on the host from the same generator
in the same order and consumes the same per-step draws: after `random.seed(s)` it starts from the reference's
initial state and a caller's own `random.randint` actions are the reference's actions, step after step (tests:
g1, g6 with no state injection).  The claim does NOT extend to `get_action_by_direction` (C-METHOD): the
reference's version draws `random.random()` / `np.random.randint` per UAV from the global generators
(uav.py:340-362) and then calls an undefined `find_closest_a_idx` (uav.py:368); the library's policy draws from
Philox, keyed by a seed derived from -- but not consuming -- the global generator's state, so a run is
reproducible under `random.seed` yet not draw-for-draw the reference's.

Speed: one `step` is ONE library call (uavtrack_step_host): the kernels write the results and the state into
a page-locked host block, the call returns after one stream synchronisation, and everything this class hands
out is read from numpy views of that block -- no per-step device allocation, no device-to-host copy calls.
"""
from __future__ import annotations

import math
import os
import random
from typing import List, Optional

import numpy as np
import torch

from .config import EnvConfig, RewardMode
from .env import BatchedUavEnv


def reference_reset_draw(config, n_uav: int, m_targets: int, x_max: float, y_max: float, action_dim: int):
    """The reference's reset (environment.py:45-107) as far as it touches Python's global `random`: the same calls in the
    same order -- per UAV `uniform(-pi, pi)` then `randint(0, action_dim - 1)` (environment.py:66-71), per target
    `uniform(0, x_max)`, `uniform(0, y_max)`, `uniform(-pi, pi)` and TARGET's unused a0 `uniform(-pi/6, pi/6)`
    (environment.py:76-81) -- and its fixed UAV layout x_i = i x_max / (n + 1), y = y_max / 2 (environment.py:105-107, from
    the CONFIG's sizes).  fp64 / int arrays: ux uy uh ua tx ty th."""
    e = config['environment']
    pi = math.pi
    n_cfg = int(e['n_uav'])
    if n_cfg < n_uav:         # the reference raises IndexError here (init_x[i] past the list, environment.py:59)
        raise ValueError(f"config['environment']['n_uav'] = {n_cfg} < the environment's {n_uav}")
    # (more UAVs in the config than in the environment: the reference lays out n_cfg positions, spaced by
    #  x_max / (n_cfg + 1), and uses the first n_uav of them -- environment.py:54-59, 105)
    init_x = [x * e['x_max'] / (n_cfg + 1) for x in range(1, n_cfg + 1)][:n_uav]
    init_y = e['y_max'] / 2
    uh, ua = [], []
    for _ in range(n_uav):
        uh.append(random.uniform(-pi, pi))
        ua.append(random.randint(0, action_dim - 1))
    tx, ty, th = [], [], []
    for _ in range(m_targets):
        tx.append(random.uniform(0, x_max))
        ty.append(random.uniform(0, y_max))
        th.append(random.uniform(-pi, pi))
        random.uniform(-pi / 6, pi / 6)
    return dict(ux=np.asarray(init_x, np.float64), uy=np.full(n_uav, init_y, np.float64), uh=np.asarray(uh, np.float64),
                ua=np.asarray(ua, np.int64), tx=np.asarray(tx, np.float64), ty=np.asarray(ty, np.float64),
                th=np.asarray(th, np.float64))


def reference_step_draws(config, m_targets: int) -> None:
    """TARGET.update_position draws one `random.uniform(-h_max, h_max)` per target per step and never uses it (target.py:34;
    the heading update below it is commented out).  Consumed so that the global generator stays in step with a reference
    run -- it is what a caller's own action draws and train.py's replay sampling read next."""
    h_max_t = math.pi / float(config["target"]["h_max"])
    for _ in range(m_targets):
        random.uniform(-h_max_t, h_max_t)


def seed_from_global_random() -> int:
    """A 63-bit seed pinned by `random.seed(...)` that does NOT advance the global generator: drawn from a copy of its
    state.  (hash(random.getstate()) would not do: the state tuple ends in None, whose hash is an address on CPython
    3.10 and changes from process to process.)"""
    r = random.Random()
    r.setstate(random.getstate())
    return r.getrandbits(63)


class _UavView:
    """What the callers touch on `env.uav_list[i]` (uav.py:12-51 attributes, :192)."""

    def __init__(self, env: "Environment", i: int):
        self._env, self._i = env, i

    def get_local_state(self) -> np.ndarray:
        return self._env._obs[self._i].copy()

    def get_action_by_direction(self, target_list=None, uav_list=None) -> int:
        """The C-METHOD baseline the way train.run_epoch calls it (train.py:350; uav.py:324-369): one call per
        UAV per step.  The library evaluates the policy for all UAVs at once (uavtrack_greedy_actions) on the
        first call of a step; the other N - 1 calls read the cached row.  The arguments are accepted for call
        compatibility -- the policy works on the environment's own state."""
        return self._env._greedy_action(self._i)

    x = property(lambda self: float(self._env._state["ux"][self._i]))
    y = property(lambda self: float(self._env._state["uy"][self._i]))
    h = property(lambda self: float(self._env._state["uh"][self._i]))
    a = property(lambda self: int(self._env._state["ua"][self._i]))
    dp = property(lambda self: self._env._cfg.dp)
    dc = property(lambda self: self._env._cfg.dc)
    # uav.raw_reward (uav.py:50, set at environment.py:219) and uav.reward (environment.py:226) of the last step.
    # (The per-UAV observation LISTS, uav.target_observation / uav.uav_communication, are scratch of the reference's
    #  get_local_state and are not materialised: the kernel folds them into the 12-d local state.)
    raw_reward = property(lambda self: float(self._env._last_raw[self._i]))
    reward = property(lambda self: float(self._env._last_reward[self._i]))


class _TargetView:
    def __init__(self, env: "Environment", k: int):
        self._env, self._k = env, k

    x = property(lambda self: float(self._env._state["tx"][self._k]))
    y = property(lambda self: float(self._env._state["ty"][self._k]))
    h = property(lambda self: float(self._env._state["th"][self._k]))


class Environment:
    def __init__(self, n_uav: int, m_targets: int, x_max: float, y_max: float, na: int,
                 device: str = "cuda:0"):
        self.x_max, self.y_max = x_max, y_max
        self.state_dim = (4 + 1) + 4 + (2 + 1)      # environment.py:28
        self.action_dim = na
        self.n_uav, self.m_targets = n_uav, m_targets
        self.uav_list: List[_UavView] = []
        self.target_list: List[_TargetView] = []
        self.position = {'all_uav_xs': [], 'all_uav_ys': [], 'all_target_xs': [], 'all_target_ys': []}
        self.covered_target_num: List[int] = []
        self._device = device
        self._env: Optional[BatchedUavEnv] = None
        self._cfg: Optional[EnvConfig] = None
        self._obs = np.zeros((n_uav, 12), dtype=np.float64)
        self._last_reward = np.zeros(n_uav)
        self._last_raw = np.zeros(n_uav)
        self._state = None                          # ux uy uh ua tx ty th of the one environment (numpy, fp32 / int32)
        self._act = np.zeros((1, n_uav), dtype=np.int32)
        self._pmi_id = None
        self._episode = 0
        self._greedy_cache = None
        self._greedy_seed: Optional[int] = None

    # -- helpers ----------------------------------------------------------------------
    def _mode_for(self, config, pmi) -> RewardMode:
        if float(config.get('cooperative', 0)) == 0:
            return RewardMode.RAW                     # uav.py:270,300
        return RewardMode.PMI if pmi else RewardMode.MEAN   # uav.py:319

    def _make(self, config, mode: RewardMode):
        cfg = EnvConfig.from_reference_dict(config, n_envs=1, reward_mode=mode, horizon=0)
        # the env object's own sizes rule the simulation; config's n_uav/m_targets only enter the reward normalisation
        # (environment.py:208-210) and, at reset, the spacing of the UAVs' start positions (environment.py:105)
        cfg = cfg.with_(n_uav=self.n_uav, m_targets=self.m_targets, x_max=float(self.x_max),
                        y_max=float(self.y_max), na=int(self.action_dim),
                        norm_n_uav=int(config['environment']['n_uav']),
                        norm_m_targets=int(config['environment']['m_targets']))
        return cfg

    def _ensure(self, config, mode: RewardMode):
        # (every value the reference reads from `config` per step or per reset, as one tuple: the dict -> EnvConfig
        #  conversion and the handle survive for as long as the caller leaves them alone)
        e, u = config['environment'], config['uav']
        key = (mode, e['n_uav'], e['m_targets'], e['x_max'], e['y_max'], e['na'], u['dt'], u['v_max'], u['h_max'], u['dc'],
               u['dp'], u['alpha'], u['beta'], u['gamma'], config['target']['v_max'], config.get('cooperative', 0))
        if self._env is not None and key == getattr(self, "_cfg_key", None):
            return
        cfg = self._make(config, mode)
        if self._env is None or cfg != self._cfg:
            state = self._env.get_state() if self._env is not None else None
            if self._env is not None:
                self._env.close()
            self._env = BatchedUavEnv(cfg, device=self._device)
            self._cfg = cfg
            self._pmi_id = None
            if state is not None:
                self._env.set_state(**state)
        self._cfg_key = key
        self._stream = self._env._stream()          # looked up once per handle: every step of this adapter ends synchronised

    def _host_state(self):
        """ux uy uh ua tx ty th of the environment as numpy arrays (kept for callers of earlier versions)."""
        return self._state

    # -- reference surface ------------------------------------------------------------
    def reset(self, config):
        """environment.py:87-107.  Returns None like the reference.

        Seed-identical with the reference: the initial state is drawn HERE, on the host, from Python's global `random`
        in the reference's own order (environment.py:54-83: per UAV `uniform(-pi, pi)` then `randint(0, na - 1)`; per target
        `uniform(0, x_max)`, `uniform(0, y_max)`, `uniform(-pi, pi)` and the unused `uniform(-pi/6, pi/6)` of TARGET's a0)
        and injected (uavtrack_set_state) -- so `random.seed(42); env.reset(cfg)` gives the reference's poses (rounded to
        the library's fp32 state) and leaves the generator where the reference leaves it."""
        self._ensure(config, self._mode_for(config, None) if self._cfg is None else self._cfg.resolved_mode())
        st0 = reference_reset_draw(config, self.n_uav, self.m_targets, self.x_max, self.y_max, self.action_dim)
        ua = st0["ua"]
        self._env.set_state(**st0, step_count=np.zeros(1, np.int32))
        self._episode += 1
        dc = float(self._cfg.dc)
        # get_states() of a fresh state: empty observation lists -> -1 (uav.py:174,186), then x / dc, y / dc, a / Na (uav.py:154)
        obs = np.full((self.n_uav, 12), -1.0)
        obs[:, 9] = st0["ux"] / dc
        obs[:, 10] = st0["uy"] / dc
        obs[:, 11] = np.asarray(ua, np.float64) / float(self.action_dim)
        self._obs = obs
        # the state as the library holds it: fp32 poses, int32 actions
        self._state = {k: (np.asarray(v, np.int32) if k == "ua" else np.asarray(v, np.float32)) for k, v in st0.items()}
        self._last_reward = np.zeros(self.n_uav)
        self._last_raw = np.zeros(self.n_uav)
        self.uav_list = [_UavView(self, i) for i in range(self.n_uav)]
        self.target_list = [_TargetView(self, k) for k in range(self.m_targets)]
        self.position = {'all_uav_xs': [], 'all_uav_ys': [], 'all_target_xs': [], 'all_target_ys': []}
        self.covered_target_num = []
        self._greedy_cache = None

    def get_states(self) -> List[np.ndarray]:
        return list(self._obs.copy())                # environment.py:109-118: one 12-vector per UAV

    def _greedy_action(self, i: int) -> int:
        step = len(self.covered_target_num)
        if self._greedy_cache is None or self._greedy_cache[0] != (self._episode, step):
            if self._greedy_seed is None:
                # pinned by random.seed like everything else, without consuming a draw (the library's policy draws are
                # Philox: the global generator stays where the environment's own draws leave it)
                self._greedy_seed = seed_from_global_random()
            acts = self._env.greedy_actions(seed=self._greedy_seed)[0].cpu().numpy()
            self._greedy_cache = ((self._episode, step), acts)
        return int(self._greedy_cache[1][i])

    def step(self, config, pmi, actions):
        """environment.py:120-164: (next_states, reward dict, covered targets).  One library call, one synchronisation."""
        mode = self._mode_for(config, pmi)
        self._ensure(config, mode)
        if mode == RewardMode.PMI and (id(pmi) != self._pmi_id or not self.covered_target_num):
            self._env.set_pmi(pmi.state_dict())   # weights change between episodes (train.py:262)
            self._pmi_id = id(pmi)
        reference_step_draws(config, self.m_targets)           # target.py:34
        self._act[0, :] = actions
        v = self._env.step_host(self._act, self._stream)       # numpy views of the library's host block
        covered = int(v["covered"][0])
        self._obs = v["obs"][0].astype(np.float64)
        self._last_reward = v["reward"][0].astype(np.float64)
        self._last_raw = v["raw"][0].astype(np.float64)
        terms = v["terms"][:, 0].astype(np.float64)
        self._state = {k: v[k][0].copy() for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")}
        st = self._state
        self.covered_target_num.append(covered)                      # environment.py:147
        self.position['all_target_xs'].append(st["tx"].tolist())     # environment.py:150-155
        self.position['all_target_ys'].append(st["ty"].tolist())
        self.position['all_uav_xs'].append(st["ux"].tolist())
        self.position['all_uav_ys'].append(st["uy"].tolist())
        reward_dict = {                                              # environment.py:157-162
            'rewards': self._last_reward.tolist(),
            'target_tracking_reward': terms[0].tolist(),
            'boundary_punishment': terms[1].tolist(),
            'duplicate_tracking_punishment': terms[2].tolist(),
        }
        return self.get_states(), reward_dict, covered

    def get_uav_and_target_position(self):
        return (self.position['all_uav_xs'], self.position['all_uav_ys'],
                self.position['all_target_xs'], self.position['all_target_ys'])   # environment.py:188-193

    def calculate_covered_target(self) -> int:
        return self.covered_target_num[-1] if self.covered_target_num else 0

    def save_position(self, save_dir, epoch_i):
        """Same CSV layout as environment.py:229-238."""
        u_xy = np.array([self.position["all_uav_xs"], self.position["all_uav_ys"]]).transpose()
        t_xy = np.array([self.position["all_target_xs"], self.position["all_target_ys"]]).transpose()
        np.savetxt(os.path.join(save_dir, "u_xy", 'u_xy' + str(epoch_i) + '.csv'),
                   u_xy.reshape(-1, 2), delimiter=',', header='x,y', comments='')
        np.savetxt(os.path.join(save_dir, "t_xy", 't_xy' + str(epoch_i) + '.csv'),
                   t_xy.reshape(-1, 2), delimiter=',', header='x,y', comments='')

    def save_covered_num(self, save_dir, epoch_i):
        """environment.py:240-244."""
        arr = np.array(self.covered_target_num).reshape(-1, 1)
        np.savetxt(os.path.join(save_dir, "covered_target_num", 'covered_target_num' + str(epoch_i) + '.csv'),
                   arr, delimiter=',', header='covered_target_num', comments='')
