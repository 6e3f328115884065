"""Environment constants.  The reference passes a YAML-derived dict into every
Environment.step call (src/environment.py:207-224, configs/*.yaml); here the same
values are captured once."""
from __future__ import annotations

import enum
import math
from dataclasses import dataclass, replace
from typing import Any, Mapping, Optional

from . import _lib


class RewardMode(enum.IntEnum):
    RAW = 0    # MAAC   (cooperative == 0: uav.py:270,300)
    MEAN = 1   # MAAC-G (pmi is None:     uav.py:293-310)
    PMI = 2    # MAAC-R (PMINetwork:      uav.py:262-291)


@dataclass(frozen=True)
class EnvConfig:
    """Defaults are the reference's configs/*.yaml `environment`/`uav`/`target` blocks."""
    n_envs: int = 1
    n_uav: int = 10
    m_targets: int = 10
    dim: int = 2
    na: int = 12
    nc: int = 1
    x_max: float = 2000.0
    y_max: float = 2000.0
    z_max: float = 500.0
    dt: float = 1.0
    u_v_max: float = 20.0
    u_h_max: float = math.pi / 6.0      # yaml `h_max: 6` means pi/6 (environment.py:100)
    u_g_max: float = math.pi / 12.0
    dc: float = 500.0
    dp: float = 200.0
    t_v_max: float = 5.0
    alpha: float = 0.6
    beta: float = 0.2
    gamma: float = 0.2
    cooperative: float = 0.0
    reward_mode: Optional[RewardMode] = None   # None: RAW if cooperative == 0 else MEAN
    horizon: int = 200                          # main.py:128 num_steps
    norm_n_uav: Optional[int] = None
    norm_m_targets: Optional[int] = None
    env_offset: int = 0

    @property
    def na_total(self) -> int:
        return self.na * self.nc

    def resolved_mode(self) -> RewardMode:
        if self.reward_mode is not None:
            return RewardMode(self.reward_mode)
        return RewardMode.RAW if self.cooperative == 0 else RewardMode.MEAN

    def clip_can_overstep(self) -> dict:
        """Which of the reference's four `clip_and_normalize` calls (environment.py:206-225) can meet a value outside its
        range with these constants -- where the reference prints "overstep in clip." (data_util.py:44-47) and the kernels
        clamp silently.  tracking: sum over m_targets of at most 2 each against the ceiling 2 * norm_m_targets; duplicate:
        n_uav - 1 terms of at least -e/2 each against the floor -e/2 * norm_n_uav; boundary: never (the raw term is in
        [-1/2, 0] by construction); reward: a weighted sum of terms in [0, 1], [-1, 0], [-1, 0] against [-1, 1]."""
        nn = self.n_uav if self.norm_n_uav is None else self.norm_n_uav
        nm = max(1, self.m_targets if self.norm_m_targets is None else self.norm_m_targets)
        hi = max(self.alpha, 0.0) + max(-self.beta, 0.0) + max(-self.gamma, 0.0)
        lo = min(self.alpha, 0.0) - max(self.beta, 0.0) - max(self.gamma, 0.0)
        return {"tracking": self.m_targets > nm, "duplicate": self.n_uav - 1 > nn, "boundary": False,
                "reward": hi > 1.0 or lo < -1.0}

    @classmethod
    def from_reference_dict(cls, config: Mapping[str, Any], n_envs: int = 1, **over) -> "EnvConfig":
        """Accepts the dict shape args_util.get_config produces (keys read at
        environment.py:97-107, 207-224)."""
        e, u, t = config["environment"], config["uav"], config["target"]
        kw = dict(n_envs=n_envs, n_uav=int(e["n_uav"]), m_targets=int(e["m_targets"]),
                  x_max=float(e["x_max"]), y_max=float(e["y_max"]), na=int(e["na"]),
                  dt=float(u["dt"]), u_v_max=float(u["v_max"]), u_h_max=math.pi / float(u["h_max"]),
                  dc=float(u["dc"]), dp=float(u["dp"]),
                  alpha=float(u["alpha"]), beta=float(u["beta"]), gamma=float(u["gamma"]),
                  t_v_max=float(t["v_max"]), cooperative=float(config.get("cooperative", 0)))
        kw.update(over)
        return cls(**kw)

    def with_(self, **kw) -> "EnvConfig":
        return replace(self, **kw)

    def c_struct(self, device_id: int) -> "_lib.UavtrackConfig":
        import ctypes as C
        c = _lib.UavtrackConfig()
        c.struct_size = C.sizeof(_lib.UavtrackConfig)
        c.n_envs, c.n_uav, c.m_targets = self.n_envs, self.n_uav, self.m_targets
        c.dim, c.na, c.nc = self.dim, self.na, self.nc
        c.norm_n_uav = self.n_uav if self.norm_n_uav is None else self.norm_n_uav
        c.norm_m_targets = max(1, self.m_targets if self.norm_m_targets is None else self.norm_m_targets)
        c.reward_mode = int(self.resolved_mode())
        c.horizon = self.horizon
        c.device_id = device_id
        c.env_offset = self.env_offset
        c.x_max, c.y_max, c.z_max = self.x_max, self.y_max, self.z_max
        c.dt, c.u_v_max, c.u_h_max, c.u_g_max = self.dt, self.u_v_max, self.u_h_max, self.u_g_max
        c.dc, c.dp, c.t_v_max = self.dc, self.dp, self.t_v_max
        c.alpha, c.beta, c.gamma = self.alpha, self.beta, self.gamma
        c.cooperative = self.cooperative
        return c
