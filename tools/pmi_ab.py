#!/usr/bin/env python3
"""A/B timing of the MAAC-R kernels (GPU box): per kernel class (library-side HIP events) for 200-step launches of the
reference shape.  usage: pmi_ab.py [--lib build_variants/X.so ...] [--hidden 128] [--launches 12]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--lib", action="append"); ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--launches", type=int, default=12); ap.add_argument("--box", type=float, default=2000.0)
a = ap.parse_args()
import subprocess
if a.lib and len(a.lib) > 1:          # one process per library (the loader caches the first one)
    for l in a.lib:
        subprocess.run([sys.executable, __file__, "--lib", l, "--hidden", str(a.hidden), "--launches", str(a.launches), "--box", str(a.box)])
    sys.exit(0)
import torch, uavtrack
from uavtrack import _lib
if a.lib and a.lib[0] != "default":
    _lib.LIB_PATH = os.path.abspath(a.lib[0])
from bench import synthetic_pmi_state_dict
B, N, M, T = 4096, 20, 10, 200
cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3, x_max=a.box, y_max=a.box, reward_mode=uavtrack.RewardMode.PMI, horizon=T)
env = uavtrack.BatchedUavEnv(cfg); env.set_pmi(synthetic_pmi_state_dict(a.hidden))
act = torch.randint(0, 12, (T, B, N), dtype=torch.int32, device="cuda", generator=torch.Generator("cuda").manual_seed(42))
out = None
for _ in range(40):                    # power state + warm-up
    env.reset(seed=42); out = env.step_many(act, out=out)
env.set_profiling(True); env.profile()
for _ in range(a.launches):
    env.reset(seed=42); out = env.step_many(act, out=out)
pr = env.profile()
print(os.path.basename(a.lib[0]) if a.lib else "default", f"H={a.hidden}",
      "  ".join(f"{k} {v['ms'] / max(v['launches'], 1):.4f} ms" for k, v in pr.items() if v["launches"]), flush=True)
