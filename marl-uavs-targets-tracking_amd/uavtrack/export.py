"""Trajectory / result export in the reference's formats, from what the batched
environment already returns: the last three observation entries are x/dc, y/dc, a/Na (uav.py:154),
so a rollout's `obs[T, B, N, 12]` carries every UAV track; the target tracks are the optional `targets[T, B, M, 2]`
output of the stepping calls (step_many(..., want_targets=True); uavtrack_set_target_trace).

    u_xy<ep>.csv            rows `x,y`: np.array([xs, ys]).transpose().reshape(-1, 2) with xs, ys of
                            shape [T][N]  ->  UAV-major, then step           (environment.py:229-238)
    t_xy<ep>.csv            the same layout for the targets                  (environment.py:232-238)
    covered_target_num<ep>.csv   one count per step                          (environment.py:240-244)
"""
from __future__ import annotations

import os

import numpy as np


def uav_tracks_from_obs(obs, dc: float, env_index: int = 0):
    """obs [T, B, N, 12] (tensor or array) -> (xs [T, N], ys [T, N]) in metres for one environment."""
    o = obs[:, env_index]
    o = o.detach().cpu().numpy() if hasattr(o, "detach") else np.asarray(o)
    return o[..., 9] * dc, o[..., 10] * dc


def save_uav_positions(save_dir: str, epoch_i, xs, ys) -> str:
    """Same file as Environment.save_position writes for the UAVs (environment.py:229-236)."""
    u_xy = np.array([np.asarray(xs), np.asarray(ys)]).transpose()          # [N, T, 2]
    path = os.path.join(save_dir, "u_xy", "u_xy" + str(epoch_i) + ".csv")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savetxt(path, u_xy.reshape(-1, 2), delimiter=",", header="x,y", comments="")
    return path


def target_tracks(targets, env_index: int = 0):
    """targets [T, B, M, 2] (tensor or array) -> (xs [T, M], ys [T, M]) for one environment."""
    t = targets[:, env_index]
    t = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return t[..., 0], t[..., 1]


def save_target_positions(save_dir: str, epoch_i, xs, ys) -> str:
    """Same file as Environment.save_position writes for the targets (environment.py:232-238)."""
    t_xy = np.array([np.asarray(xs), np.asarray(ys)]).transpose()          # [M, T, 2]
    path = os.path.join(save_dir, "t_xy", "t_xy" + str(epoch_i) + ".csv")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savetxt(path, t_xy.reshape(-1, 2), delimiter=",", header="x,y", comments="")
    return path


def save_rollout(save_dir: str, epoch_i, result, dc: float, env_index: int = 0):
    """Everything Environment.save_position + save_covered_num write for one environment of a batched rollout
    (`result` = step_many / run_actor / run_greedy output with want_targets=True).  Returns the three paths."""
    xs, ys = uav_tracks_from_obs(result["obs"], dc, env_index)
    paths = [save_uav_positions(save_dir, epoch_i, xs, ys)]
    txs, tys = target_tracks(result["targets"], env_index)
    paths.append(save_target_positions(save_dir, epoch_i, txs, tys))
    paths.append(save_covered_num(save_dir, epoch_i, result["covered"], env_index))
    return paths


def save_covered_num(save_dir: str, epoch_i, covered, env_index: int = 0) -> str:
    """covered [T, B] -> the file of Environment.save_covered_num (environment.py:240-244)."""
    c = covered[:, env_index]
    c = c.detach().cpu().numpy() if hasattr(c, "detach") else np.asarray(c)
    path = os.path.join(save_dir, "covered_target_num", "covered_target_num" + str(epoch_i) + ".csv")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savetxt(path, c.reshape(-1, 1), delimiter=",", header="covered_target_num", comments="")
    return path
