#!/usr/bin/env python3
"""Diagnostic (GPU box): per-phase s_memtime totals of the stamped step-kernel build (ep_sums carries the cycle sums)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
lib, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
import torch, uavtrack
from uavtrack import _lib
_lib.LIB_PATH = os.path.abspath(lib)
cfg = uavtrack.EnvConfig(n_envs=B, n_uav=20, m_targets=10)
env = uavtrack.BatchedUavEnv(cfg)
env.reset(seed=1)
act = torch.randint(0, 12, (T, B, 20), dtype=torch.int32, device="cuda")
out = env.step_many(act)
env.reset(seed=1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); out = env.step_many(act, out=out); e1.record(); torch.cuda.synchronize()
ep = out["ep_sums"].double().cpu()
names = ["P1a targets", "P1b kinematics+LDS write+barrier", "P2 sweeps", "P3 finalize+barrier", "P4 outputs"]
print(f"{os.path.basename(lib)} B={B} T={T}: {e0.elapsed_time(e1):.3f} ms; s_memtime ticks per step (mean over envs / min / max):")
tot = 0
for k, n in enumerate(names):
    v = ep[:, k] / T
    tot += v.mean().item()
    print(f"  {n:36s} {v.mean().item():8.1f} {v.min().item():8.1f} {v.max().item():8.1f}")
print(f"  total {tot:.1f} ticks/step = {tot / 100e6 * 1e6:.3f} us at 100 MHz")
