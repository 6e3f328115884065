#!/usr/bin/env python3
"""Experiment (GPU box): the PCIe-inclusive rate of the host-facing step (uavtrack_step_host) at batch sizes beyond the
adapter's B = 1 -- every output of the step crosses PCIe into the page-locked block."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
import numpy as np, torch, uavtrack
for B in (1, 64, 1024, 4096):
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=B, n_uav=20, m_targets=10))
    env.reset(seed=1)
    act = np.random.RandomState(0).randint(0, 12, size=(B, 20)).astype(np.int32)
    for _ in range(20):
        env.step_host(act)
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        env.step_host(act)
    dt = (time.perf_counter() - t0) / n
    nbytes = B * 20 * (48 + 4 + 12 + 4 + 4) + B * 5 + B * (20 * 16 + 10 * 12 + 8)
    print(f"B={B:5d}: {dt * 1e6:8.1f} us per host step = {B * 20 / dt / 1e6:8.2f} M agent-steps/s, {nbytes / dt / 1e9:6.2f} GB/s over PCIe ({nbytes / 1e6:.2f} MB per step)")
    env.close()
