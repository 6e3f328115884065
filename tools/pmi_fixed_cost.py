#!/usr/bin/env python3
"""Fixed cost of the MAAC-R per-step launches (GPU box): a box so large that no pair is ever in range."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
import torch, uavtrack
sys.path.insert(0, ROOT)
from bench import synthetic_pmi_state_dict
for box in (1e6, 2000.0):
    cfg = uavtrack.EnvConfig(n_envs=4096, n_uav=20, m_targets=10, cooperative=0.3, x_max=box, y_max=box,
                             reward_mode=uavtrack.RewardMode.PMI)
    env = uavtrack.BatchedUavEnv(cfg); env.set_pmi(synthetic_pmi_state_dict(128)); env.reset(seed=1)
    act = torch.randint(0, 12, (1, 4096, 20), dtype=torch.int32, device="cuda")
    for _ in range(150): out = env.step_many(act)          # let the line-up disperse
    torch.cuda.synchronize(); p0 = env.pmi_pairs_scored()
    t0 = time.perf_counter()
    for _ in range(200): out = env.step_many(act, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"box {box:g}: {dt/200*1e6:.1f} us per step, {(env.pmi_pairs_scored()-p0)/200:.0f} pairs per step", flush=True)
