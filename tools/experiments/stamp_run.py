#!/usr/bin/env python3
"""Experiment (GPU box): in-kernel time stamps of short rollout launches (build_variants/stamp.so from stamp_build.py).
usage: stamp_run.py [B]   -- 20 x 10 MAAC, T in (1, 5, 20)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
os.environ["UAVTRACK_LIB"] = os.path.join(ROOT, "build_variants", "stamp.so")
import numpy as np, torch, uavtrack
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = uavtrack.EnvConfig(n_envs=B, n_uav=20, m_targets=10)
env = uavtrack.BatchedUavEnv(cfg)
info = env.kernel_info()
G, E = info["workgroups"], info["envs_per_workgroup"]
assert E == 3, info
print(f"B={B}: {G} workgroups of {info['workgroup']} threads, {E} envs each")
for T in (1, 5, 20):
    act = torch.randint(0, 12, (T, B, 20), dtype=torch.int32, device="cuda")
    env.reset(seed=1)
    out = env.step_many(act)
    rows = []
    for rep in range(40):
        env.reset(seed=1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # a few launches back to back: the middle one is measured (no idle GPU in front of it)
        out = env.step_many(act, out=out)
        e0.record(); out = env.step_many(act, out=out); e1.record()
        torch.cuda.synchronize()
        G1 = G - 1             # (the last workgroup may hold fewer than 3 environments: its 15 words are not all there)
        st = out["ep_sums"].view(torch.int32).reshape(-1)[: 15 * G1].reshape(G1, 15)[:, :4].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        t0, t1, t2, t3 = (st[:, k] * 10.0 for k in range(4))          # ns (100 MHz)
        base = t0.min()
        rows.append([e0.elapsed_time(e1) * 1e3, (t3.max() - base) / 1e3, (t0.max() - base) / 1e3, np.median(t1 - t0) / 1e3,
                     np.median(t2 - t1) / 1e3, np.median(t3 - t2) / 1e3, np.median(t3 - t0) / 1e3, (np.sort(t3)[G1 // 2] - base) / 1e3])
    r = np.median(np.array(rows), axis=0)
    print(f"T={T:2d}: events {r[0]:.2f} us | first entry -> last exit {r[1]:.2f} | last workgroup enters at {r[2]:.2f} | per workgroup (median): "
          f"load+fill {r[3]:.2f}, steps {r[4]:.2f} ({r[4] / T:.2f}/step), tail {r[5]:.2f}, entry->exit {r[6]:.2f}; median exit at {r[7]:.2f}")
