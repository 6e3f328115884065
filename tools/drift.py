#!/usr/bin/env python3
"""Why do back-to-back 200-step launches of the headline workload drift (0.577 -> 0.612 ms over six launches
after a short warm-up, steady 0.546-0.561 after a long one)?  Times series of identical launches (HIP events on the launch
stream, everything enqueued before the first wait) under different histories of the SAME process:
  A  fresh output buffers, 1 untimed launch            (what bench.py's roofline leg did in round 2)
  B  the same buffers again, after 0.5 s of idle
  C  after 300 back-to-back launches (~170 ms of load), no idle gap
  D  after C plus 0.5 s idle
  E  fresh buffers of a second handle, right after C-like load
  F  series of 60 with NO reset between launches
Prints one JSON line per series.  usage: drift.py [--envs 4096] [--n 40]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--n", type=int, default=40); ap.add_argument("--T", type=int, default=200)
a = ap.parse_args()
import torch
import uavtrack

dev = "cuda:0"
N, M, T, B = 20, 10, a.T, a.envs


def make():
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, horizon=T), dev)
    act = torch.randint(0, 12, (T, B, N), dtype=torch.int32, device=dev, generator=torch.Generator(dev).manual_seed(42))
    out = None
    return env, act, out


def series(env, act, out, n, reset=True):
    ev = []
    for _ in range(n):
        if reset:
            env.reset(seed=42)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = env.step_many(act, out=out); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    return [round(x.elapsed_time(y), 4) for x, y in ev], out


def report(tag, ms):
    s = sorted(ms)
    print(json.dumps({"series": tag, "n": len(ms), "median": s[len(s) // 2], "min": s[0], "max": s[-1],
                      "first6": ms[:6], "last6": ms[-6:], "all": ms}), flush=True)


env, act, out = make()
env.reset(seed=42); out = env.step_many(act); torch.cuda.synchronize()
ms, out = series(env, act, out, a.n); report("A fresh buffers, 1 untimed launch", ms)
time.sleep(0.5)
ms, out = series(env, act, out, a.n); report("B same buffers after 0.5 s idle", ms)
_, out = series(env, act, out, 300)
ms, out = series(env, act, out, a.n); report("C right after 300 launches", ms)
time.sleep(0.5)
ms, out = series(env, act, out, a.n); report("D after C + 0.5 s idle", ms)
_, out = series(env, act, out, 300)
env2, act2, out2 = make()
env2.reset(seed=42); out2 = env2.step_many(act2); torch.cuda.synchronize()
ms, out2 = series(env2, act2, out2, a.n); report("E second handle, fresh buffers, after load", ms)
ms, out2 = series(env2, act2, out2, 60, reset=False); report("F no reset between launches", ms)
# G: wait for each launch before issuing the next (host in the loop)
msg = []
for _ in range(a.n):
    env2.reset(seed=42)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out2 = env2.step_many(act2, out=out2); e1.record(); torch.cuda.synchronize()
    msg.append(round(e0.elapsed_time(e1), 4))
report("G synchronize after every launch", msg)
