"""Where a short timed region goes: wall of (launch + device sync) for a T-step uavtrack_step_many against the
kernel's own HIP-event time.  Usage: python3 tools/launch_overhead.py [--spin] [--T 20]"""
import argparse, ctypes, os, sys, time, statistics
ap = argparse.ArgumentParser()
ap.add_argument("--spin", action="store_true")
ap.add_argument("--T", type=int, default=20)
ap.add_argument("--envs", type=int, default=4096)
a = ap.parse_args()
if a.spin:
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipSetDeviceFlags(ctypes.c_uint(1))   # hipDeviceScheduleSpin
    print("hipSetDeviceFlags(spin) rc", rc)
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "marl-uavs-targets-tracking_amd"))
import uavtrack
dev = torch.device("cuda:0")
cfg = uavtrack.EnvConfig(n_envs=a.envs, n_uav=20, m_targets=10)
env = uavtrack.BatchedUavEnv(cfg, device=dev)
T = a.T
acts = torch.randint(0, 12, (T, a.envs, 20), dtype=torch.int32, device=dev)
out = dict(obs=torch.empty(T, a.envs, 20, 12, device=dev), reward=torch.empty(T, a.envs, 20, device=dev),
           terms=torch.empty(T, 3, a.envs, 20, device=dev), covered=torch.empty(T, a.envs, dtype=torch.int32, device=dev),
           done=torch.empty(T, a.envs, dtype=torch.uint8, device=dev), ep_sums=torch.zeros(a.envs, 5, device=dev))
call = env.bind_step_many(acts, out)
env.reset(seed=1)
for _ in range(3):
    call()
torch.cuda.synchronize()
walls, kern, enq = [], [], []
for r in range(60):
    env.reset(seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    walls.append((t2 - t0) * 1e6); enq.append((t1 - t0) * 1e6)
    env.reset(seed=1)
    torch.cuda.synchronize()
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    kern.append(e0.elapsed_time(e1) * 1e3)
med = statistics.median
print(f"spin={a.spin} T={T} B={a.envs}: wall launch+sync median {med(walls):.1f} us (min {min(walls):.1f}), enqueue {med(enq):.1f} us, "
      f"kernel (events) {med(kern):.1f} us -> {a.envs*20*T/med(walls)/1e3:.2f} G agent-steps/s in the timed region")
