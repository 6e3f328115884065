#!/usr/bin/env python3
"""Experiment (GPU box): can a MAAC-R rollout and the MFMA pair scorer of the previous chunk share the chip?

The scorer (pmi_score_t3_kernel<128>: 4 wavefronts per CU, 476 of the 512 registers of every SIMD lane, 107 KB of LDS)
and the rollout (single-wavefront groups, 256 registers) cannot be co-resident on one SIMD, so overlap can only come from
giving the two kernels DIFFERENT compute units (hipExtStreamCreateWithCUMask).  This script times
  serial        scorer, then rollout, one stream
  two streams   the same two launches on two ordinary streams (what the hardware scheduler makes of it)
  k / 256-k     the scorer on the first k CUs of the mask order, the rollout on the other 256 - k
for one rollout launch of T steps and one scorer launch over n pairs (defaults: half a 200-step episode of the 4096 x 20 x 10
MAAC-R workload each).  usage: overlap_probe.py [T] [pairs] [hidden]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
import torch          # noqa: E402
import uavtrack       # noqa: E402
import bench          # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100
NPAIRS = int(sys.argv[2]) if len(sys.argv) > 2 else 1_190_000
H = int(sys.argv[3]) if len(sys.argv) > 3 else 128
B, N, M = 4096, 20, 10
dev = "cuda:0"
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def masked_stream(bits):
    """hipStream_t restricted to the CUs whose bit is set (mask order: the runtime's; on a multi-XCD part consecutive bits
    go round the XCDs)."""
    words = (C.c_uint32 * ((NCU + 31) // 32))()
    for k in bits:
        words[k // 32] |= 1 << (k % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


def make_env(reward):
    pmi = reward == uavtrack.RewardMode.PMI
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3 if pmi else 0.0, reward_mode=reward, horizon=200)
    env = uavtrack.BatchedUavEnv(cfg, dev)
    if pmi:
        env.set_pmi(bench.synthetic_pmi_state_dict(H, 42))
    return env


scorer_env = make_env(uavtrack.RewardMode.PMI)
# (the MAAC rollout kernel stands in for the MAAC-R one: the same single-wavefront geometry and instruction mix without the
#  pair emission -- a MAAC-R step_many call would bring its own scorer and mix launches into the measurement)
roll_env = make_env(uavtrack.RewardMode.RAW)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.rand(NPAIRS, 12, device=dev, generator=g) * 2 - 1
scores = torch.empty(NPAIRS, device=dev)
act = torch.randint(0, 12, (T, B, N), dtype=torch.int32, device=dev, generator=g)
out = None


def scorer():
    scorer_env.pmi_inference(x, out=scores)


def rollout():
    global out
    roll_env.reset(seed=3)
    out = roll_env.step_many(act, out=out)


def timed(fn_a, st_a, fn_b, st_b, reps=12):
    """fn_a on st_a and fn_b on st_b, both released together behind one event; wall = both done."""
    main = torch.cuda.current_stream()
    ts = []
    for r in range(reps + 3):
        torch.cuda.synchronize()
        e0, ea, eb = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record(main)
        st_a.wait_event(e0); st_b.wait_event(e0)
        with torch.cuda.stream(st_a):
            fn_a(); ea.record(st_a)
        with torch.cuda.stream(st_b):
            fn_b(); eb.record(st_b)
        torch.cuda.synchronize()
        if r >= 3:
            ts.append((e0.elapsed_time(ea), e0.elapsed_time(eb)))
    a = sorted(t[0] for t in ts)[len(ts) // 2]
    b = sorted(t[1] for t in ts)[len(ts) // 2]
    w = sorted(max(t) for t in ts)[len(ts) // 2]
    return a, b, w


# warm everything (allocations, clocks)
for _ in range(5):
    scorer(); rollout()
torch.cuda.synchronize()
s1 = torch.cuda.Stream()
s2 = torch.cuda.Stream()
print(f"T={T} steps of {B}x{N}x{M} MAAC rollout, {NPAIRS} pairs through the H={H} scorer, {NCU} CUs")
sa, _, _ = timed(scorer, s1, lambda: None, s2)
_, rb, _ = timed(lambda: None, s1, rollout, s2)
print(f"alone: scorer {sa:.3f} ms, rollout call {rb:.3f} ms, sum {sa + rb:.3f}")


def both_one_stream():
    scorer(); rollout()


w1, _, _ = timed(both_one_stream, s1, lambda: None, s2)
print(f"serial on one stream: {w1:.3f} ms")
a, b, w = timed(scorer, s1, rollout, s2)
print(f"two ordinary streams: scorer done {a:.3f}, rollout done {b:.3f}, both {w:.3f} ms")
for k in (224, 192, 160, 128, 96, 64):
    sa_ = masked_stream(range(0, k))
    sb_ = masked_stream(range(k, NCU))
    a, b, w = timed(scorer, sa_, rollout, sb_)
    a1, _, _ = timed(scorer, sa_, lambda: None, sb_)
    _, b1, _ = timed(lambda: None, sa_, rollout, sb_)
    print(f"scorer on {k:3d} CUs / rollout on {NCU - k:3d}: together scorer {a:.3f} rollout {b:.3f} both {w:.3f} ms   "
          f"(alone on their masks: {a1:.3f} / {b1:.3f})")
