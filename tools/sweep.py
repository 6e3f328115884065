#!/usr/bin/env python3
"""A/B timing helper (GPU box): times uavtrack_step_many for one config under several
workgroup sizes / library builds.  usage: sweep.py [--lib path.so] [--envs B] [--T T] [--wgs 64,128,...]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--lib"); ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--T", type=int, default=200)
ap.add_argument("--n", type=int, default=20); ap.add_argument("--m", type=int, default=10)
ap.add_argument("--coop", type=float, default=0.0); ap.add_argument("--wgs", default="0")
ap.add_argument("--policy", default="given", choices=["given", "actor", "actor_step", "greedy"]); ap.add_argument("--hidden", type=int, default=128)
ap.add_argument("--warm", type=int, default=0, help="untimed launches ahead of the timed ones, no idle gap (loaded clocks: ~150 for 80 ms)");
ap.add_argument("--reps", type=int, default=5); ap.add_argument("--reset", action="store_true", help="an episode per launch: reset ahead of every launch (what bench.py does)"); ap.add_argument("--no-obs", action="store_true"); ap.add_argument("--no-terms", action="store_true"); ap.add_argument("--dim", type=int, default=2)
a = ap.parse_args()
import torch
import uavtrack
from uavtrack import _lib
if a.lib:
    _lib.LIB_PATH = os.path.abspath(a.lib)
for wgs in [int(w) for w in a.wgs.split(",")]:
    if wgs: os.environ["UAVTRACK_WGS"] = str(wgs)
    else: os.environ.pop("UAVTRACK_WGS", None)
    cfg = uavtrack.EnvConfig(n_envs=a.envs, n_uav=a.n, m_targets=a.m, cooperative=a.coop, dim=a.dim, nc=3 if a.dim == 3 else 1)
    env = uavtrack.BatchedUavEnv(cfg)
    env.reset(seed=1)
    act = torch.randint(0, cfg.na_total, (a.T, a.envs, a.n), dtype=torch.int32, device="cuda")
    obs0 = env.reset(seed=1).clone()
    if a.policy.startswith("actor"):
        torch.manual_seed(0)
        env.set_actor(uavtrack.ActorMLP(hidden_dim=a.hidden, action_dim=cfg.na_total))

    def once(out):
        if a.reset: env.reset(seed=1)
        if a.policy == "given":
            return env.step_many(act, out=out, want_obs=not a.no_obs, want_terms=not a.no_terms)
        if a.policy == "actor":
            return env.run_actor(a.T, obs0, seed=3, want_terms=not a.no_terms, out=out)
        if a.policy == "actor_step":      # the stand-alone policy kernel alone, T launches
            for _ in range(a.T):
                out = env.actor_actions(obs0, seed=3, out=out)
            return out
        return env.run_greedy(a.T, seed=3)
    out = once(None)
    torch.cuda.synchronize()
    best = []
    if a.warm:                      # everything enqueued before the first wait: the timed launches run at loaded clocks
        for _ in range(a.warm): out = once(out)
        ev = []
        for r in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); out = once(out); e1.record(); ev.append((e0, e1))
        torch.cuda.synchronize()
        best = [x.elapsed_time(y) for x, y in ev]
    for r in range(0 if a.warm else a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = once(out); e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1))
    ms = sorted(best)[len(best) // 2]
    rate = a.envs * a.n * a.T / (ms * 1e-3)
    print(f"lib={os.path.basename(a.lib or 'default')} policy={a.policy} B={a.envs} N={a.n} M={a.m} T={a.T} wgs={env.kernel_info()['workgroup']} "
          f"median {ms:.3f} ms  min {min(best):.3f} ms  max {max(best):.3f} ms  {rate/1e9:.2f} G agent-steps/s", flush=True)
    env.close()
