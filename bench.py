#!/usr/bin/env python3
"""bench.py -- headline benchmark of the environment hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one Environment.step over the whole batch (every env advances once).
At N = 1 the workload is BASELINE.json configs[1]: 4096 envs x 20 UAVs x 10 targets,
2-D, MAAC tracking reward; for N > 1 every rank runs that same batch on its own GPU
(weak scaling, configs[4]) and the only collective is the end-of-rollout all-gather of
the per-env episode accumulators over RCCL.

`--gpus N` with N > 1 starts the N ranks itself (one child process per GPU, spawned before
the parent makes any GPU call; the parent only waits); under torchrun (WORLD_SIZE set) the
process is a rank already.  If N ranks cannot be had the run exits non-zero -- it never
degrades to fewer GPUs.

Timed region (`value`, `ms_per_step`): EXACTLY K steps issued as fused rollouts of at most
`--rollout` (default 200 = the reference horizon, main.py:128) steps per launch, never across
an episode end, with pre-sampled int32 actions resident in HBM and a reset at
every episode end, bracketed by barrier + synchronize, max over ranks.  `config.launch`
names the launches that were really timed.

`roofline` does not depend on K: rank 0 times a fixed leg of 200-step rollout launches
(the reference horizon), all enqueued back to back before the first event is waited on, so host
latency is off the clock; HIP events on the launch stream around every launch.
`cpu_baseline` is the C oracle (a port of the reference algorithm) on this host's cores.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_CEILING_GBS = 6290.0
ROOFLINE_T = 200               # steps per launch of the roofline leg: the reference horizon (main.py:128)
ROOFLINE_LAUNCHES = 10         # timed launches of a roofline leg (median / min / max are reported beside the average)
# An MI355X that has idled (even 0.5 s) runs its next launches 10-20 % slow and settles over ~20 ms of continuous load
# (tools/drift.py: 0.60, 0.56, 0.56, 0.61, 0.62, 0.62 ... 0.51 ms over 35 back-to-back launches of the headline rollout,
# 0.51 flat right after sustained load, whatever the buffers): the power state ramps.  Every measured leg is therefore
# preceded, without an idle gap, by this much untimed work of the same kind.
DEVICE_WARM_MS = 60.0
BF16_MFMA_PEAK_TFLOPS = 2500.0
FP32_MFMA_PEAK_TFLOPS = 157.3


def algorithmic_bytes_per_agent_step(n_uav, m_targets, dim=2, terms=True):
    """read UAV x,y,h,a_prev + action 20, write UAV state 16, obs 48, reward 4,
    3 reward terms 12, targets r+w 24*M/N, covered+done 5/N  (3-D: +8 and 32*M/N)."""
    b = 20 + 16 + 48 + 4 + (12 if terms else 0) + 5.0 / n_uav
    if dim == 3:
        return b + 8 + 32.0 * m_targets / n_uav
    return b + 24.0 * m_targets / n_uav


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--n-uav", type=int, default=20)
    ap.add_argument("--m-targets", type=int, default=10)
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--reward", choices=["raw", "mean", "pmi"], default="raw",
                    help="raw = MAAC (configs[1]), mean = MAAC-G, pmi = MAAC-R (configs[2])")
    ap.add_argument("--pmi-hidden", type=int, default=128)
    ap.add_argument("--pmi-scheme", choices=["auto", "f16x3", "bf16x6", "fp32"], default="auto",
                    help="pin the MAAC-R pair scorer (uavtrack_set_pmi_scheme); auto = the fastest the weights allow")
    ap.add_argument("--verbose", action="store_true", help="keep the per-launch arrays and the prose notes on the line")
    ap.add_argument("--gather-transitions", type=int, default=0,
                    help="transitions every rank samples per rollout for the learner-side all-gather (SURVEY 8e's optional second "
                         "exchange; 0 = the end-of-rollout summary gather only, which is what north_star names).  The sampling runs at "
                         "ANY world size, so the N = 1 line carries the same per-rollout work as the N > 1 lines")
    ap.add_argument("--rollout", type=int, default=200, help="steps per fused launch (1 = one launch per step)")
    ap.add_argument("--policy", choices=["given", "greedy", "actor"], default="given",
                    help="where actions come from: pre-sampled (the headline workload), the fused greedy baseline "
                         "(uavtrack_run_greedy) or the fused FnnPolicyNet actor (uavtrack_run_actor, hidden --actor-hidden)")
    ap.add_argument("--actor-hidden", type=int, default=128)
    ap.add_argument("--box", type=float, default=2000.0,
                    help="side of the square field in metres (reference: 2000; 500 = the dense MAAC-R worst case of SURVEY 8d)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip per-step-launch, saturating-batch and closed-loop legs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-device-warmup", action="store_true",
                    help="skip the untimed device warm-up ahead of the timed region (the GPU then starts from its idle power state)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the roofline legs of BASELINE configs[2] and configs[3]")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo only to rehearse "
                    "the multi-rank code path on a single GPU together with UAVTRACK_BENCH_ONE_GPU=1")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="launcher check without a GPU: the ranks rendezvous, all-reduce one number and rank 0 prints a "
                         "line with n_gpus / rccl_world_size (used by the CPU test of the --gpus N spawn path)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# --gpus N: the launcher.  Runs in a parent that never touches the GPU.
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args, argv):
    """Start args.gpus copies of this script, one per GPU, with the torchrun environment contract
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Rank 0 inherits stdout (its JSON line is
    the run's output); the other ranks' stdout goes to stderr.  Returns the exit code for the parent."""
    n = args.gpus
    if not args.selftest_launcher and args.backend == "nccl" and os.environ.get("UAVTRACK_BENCH_ONE_GPU") != "1":
        import torch                                  # device_count() does not initialise the GPU
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) are visible; refusing to run on fewer\n")
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # like torchrun: one OpenMP thread per rank unless the caller says otherwise.  os.cpu_count() reports the whole host
        # (256 on the GPU boxes) while the cgroup grants 16 cores: N ranks each spinning a 128-thread pool starve the
        # collective's own threads (the gloo rehearsal: 200 ms per 458 KB gather)
        env.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = None
    while procs:
        for p in list(procs):
            code = p.poll()
            if code is None:
                continue
            procs.remove(p)
            if code != 0 and rc == 0:
                rc = code
                deadline = time.time() + 30          # a rank died: give the others a moment, then stop them
        if deadline is not None and time.time() > deadline:
            for p in procs:
                p.kill()
        time.sleep(0.05)
    if rc != 0:
        sys.stderr.write(f"bench.py: a rank exited with code {rc}; --gpus {n} could not be honoured\n")
    return rc


def launcher_selftest(args, world, rank):
    """No GPU: proves that the spawn path yields `world` ranks that can talk."""
    import torch
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(args.backend if args.backend != "nccl" else "gloo")
        t = torch.tensor([float(rank + 1)])
        dist.all_reduce(t)
        assert int(t.item()) == world * (world + 1) // 2
        ws = dist.get_world_size()
    else:
        ws = 1
    if rank == 0:
        # the timed region's launches and gathers as time_config() would issue them with these flags (pure host logic)
        warm_plan, pos = launch_plan(args.warmup, args.rollout, ROOFLINE_T, 0)
        timed_plan, _ = launch_plan(args.steps, args.rollout, ROOFLINE_T, pos)
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "rccl_world_size": ws, "gpus_requested": args.gpus,
                          "timed_launch_steps": timed_plan,
                          "gather": {"in_region": len(gather_points(timed_plan, pos, ROOFLINE_T)),
                                     "behind_launches": gather_points(timed_plan, pos, ROOFLINE_T)}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------
def synthetic_pmi_state_dict(hidden, seed=42):
    """Random-init PMINetwork-shaped weights (PMINet.py:29-38 shapes, torch default init ranges)
    with non-trivial BatchNorm statistics; there is no checkpoint to load."""
    import numpy as np
    r = np.random.RandomState(seed)
    sd = {}
    for lin, bn, fan_in in (("fc_comm", "bn_comm", 5), ("fc_obs", "bn_obs", 4),
                            ("fc_boundary_state", "bn_boundary_state", 3), ("fc1", "bn1", 3 * hidden)):
        k = 1.0 / np.sqrt(fan_in)
        sd[lin + ".weight"] = r.uniform(-k, k, (hidden, fan_in)).astype(np.float32)
        sd[lin + ".bias"] = r.uniform(-k, k, hidden).astype(np.float32)
        sd[bn + ".weight"] = r.uniform(0.5, 1.5, hidden).astype(np.float32)
        sd[bn + ".bias"] = (r.randn(hidden) * 0.2).astype(np.float32)
        sd[bn + ".running_mean"] = (r.randn(hidden) * 0.3).astype(np.float32)
        sd[bn + ".running_var"] = r.uniform(0.5, 2.0, hidden).astype(np.float32)
    k = 1.0 / np.sqrt(hidden)
    sd["fc2.weight"] = r.uniform(-k, k, (1, hidden)).astype(np.float32)
    sd["fc2.bias"] = r.uniform(-k, k, 1).astype(np.float32)
    return sd


def make_env(uavtrack, args, B, device, env_offset=0):
    nc = 3 if args.dim == 3 else 1
    mode = {"raw": uavtrack.RewardMode.RAW, "mean": uavtrack.RewardMode.MEAN, "pmi": uavtrack.RewardMode.PMI}[args.reward]
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=args.n_uav, m_targets=args.m_targets, dim=args.dim,
                             x_max=getattr(args, "box", 2000.0), y_max=getattr(args, "box", 2000.0), nc=nc,
                             cooperative=args.cooperative, reward_mode=mode,
                             horizon=ROOFLINE_T, env_offset=env_offset)
    env = uavtrack.BatchedUavEnv(cfg, device)
    if args.reward == "pmi":
        env.set_pmi_scheme(getattr(args, "pmi_scheme", "auto"))
        env.set_pmi(synthetic_pmi_state_dict(args.pmi_hidden, 42))
    if getattr(args, "policy", "given") == "actor":
        import torch
        torch.manual_seed(args.seed)
        env.set_actor(uavtrack.ActorMLP(hidden_dim=args.actor_hidden, action_dim=env.cfg.na_total))
    return env


def launch_plan(steps, rollout, horizon, ep_steps):
    """Launch sizes for `steps` env steps: at most `rollout` steps per launch, never across an episode
    end.  Returns ([T, ...], episode position afterwards)."""
    plan = []
    while steps > 0:
        T = min(rollout, steps, horizon - ep_steps)
        plan.append(T)
        steps -= T
        ep_steps = (ep_steps + T) % horizon
    return plan, ep_steps


def gather_points(plan, ep_steps, horizon):
    """Indices of the launches of `plan` behind which the end-of-rollout gather runs: every launch that ends an episode,
    and the LAST launch whatever it ends on -- the K steps of a timed region are a rollout, and its summaries are gathered
    before the clock stops (at the driver's --steps 20 that is exactly one gather, behind the one launch)."""
    pts = []
    for k, T in enumerate(plan):
        ep_steps += T
        if ep_steps >= horizon:
            pts.append(k)
            ep_steps = 0
    if plan and (not pts or pts[-1] != len(plan) - 1):
        pts.append(len(plan) - 1)
    return pts


def describe_plan(plan):
    """'3 x 200 + 1 x 20' -- the launches of a plan, in order, run-length encoded."""
    runs = []
    for T in plan:
        if runs and runs[-1][1] == T:
            runs[-1][0] += 1
        else:
            runs.append([1, T])
    return " + ".join(f"{n} x {T}" for n, T in runs) + " steps"


def alloc_outputs(args, B, T, device):
    import torch
    import uavtrack._lib as _l
    out = dict(obs=torch.empty(T, B, args.n_uav, _l.OBS_DIM, device=device),
               reward=torch.empty(T, B, args.n_uav, device=device),
               terms=torch.empty(T, 3, B, args.n_uav, device=device),
               covered=torch.empty(T, B, dtype=torch.int32, device=device),
               done=torch.empty(T, B, dtype=torch.uint8, device=device),
               ep_sums=torch.empty(B, 5, device=device))
    if args.policy == "actor":
        out["actions"] = torch.empty(T, B, args.n_uav, dtype=torch.int32, device=device)
    return out


def run_rollouts(env, actions, plan, ep_steps, out, gather=None, policy="given", obs=None, bound=None):
    """Issue the launches of `plan`; gathers the rollout summaries at gather_points() and resets at every episode end.
    obs: the observation the in-kernel policy sees first (policy != "given").
    bound: pre-built launch callables from bind_plan (the timed region replays those: no per-call Python beyond the
    ctypes call itself)."""
    horizon = env.cfg.horizon
    gather_at = set(gather_points(plan, ep_steps, horizon)) if gather is not None else set()
    for k, T in enumerate(plan):
        if bound is not None:
            res = bound["launch"][k]()
        elif policy == "actor":
            res = env.run_actor(T, obs, seed=42, out=out.get(T))
            obs = res["obs"][-1]
        elif policy == "greedy":
            res = env.run_greedy(T, seed=42)
        else:
            res = env.step_many(actions[ep_steps:ep_steps + T], out=out.get(T))   # the episode's own action rows
        out[T] = res
        ep_steps += T
        if k in gather_at:                    # end of a rollout: gather its summaries (and sampled transitions)
            gather(res, None if actions is None else actions[ep_steps - T:ep_steps])
        if ep_steps >= horizon:               # end of an episode: start the next one
            if bound is not None:
                bound["episode"] += 1
                obs = bound["reset"](bound["episode"])
            else:
                obs = env.reset(seed=42)
            ep_steps = 0
    return len(plan), obs


def bind_plan(env, actions, plan, ep_steps, out, device):
    """Pre-built launch callables for `plan` (pre-sampled actions only): the timed region then issues exactly the
    library calls, nothing else."""
    import torch
    import uavtrack._lib as _l
    horizon = env.cfg.horizon
    launches = []
    for T in plan:
        launches.append(env.bind_step_many(actions[ep_steps:ep_steps + T], out[T]))
        ep_steps = (ep_steps + T) % horizon
    obs = torch.empty(env.B, env.N, _l.OBS_DIM, device=device)
    return dict(launch=launches, reset=env.bind_reset(42, obs), episode=1 << 20)


def time_config(uavtrack, args, B, steps, warmup, rollout, device, dist=None, env_offset=0, total_envs=None):
    """The contract's timed region: `warmup` untimed steps, then exactly `steps` steps between
    barrier + synchronize.  Returns wall time, the plan that was timed and per-launch HIP-event times."""
    import torch
    total_envs = total_envs or B
    env = make_env(uavtrack, args, B, device, env_offset)
    na_total = env.cfg.na_total
    g = torch.Generator(device=device).manual_seed(args.seed + env_offset)
    horizon = env.cfg.horizon
    actions = torch.randint(0, na_total, (horizon, B, args.n_uav), dtype=torch.int32, device=device, generator=g)
    multi = dist is not None and dist.is_initialized() and dist.get_world_size() > 1
    gathered_bytes = [0, 0]          # per rank: episode summaries, sampled transitions
    gathers = [0]                    # end-of-rollout gathers issued so far
    # The end-of-rollout exchange.  The SAME calls run at every world size: without a process group the two
    # gather functions hand their input back (no copy, no collective), so the N = 1 line and the N > 1 lines differ by the
    # collective and nothing else -- whatever host and stream work rides with it (the optional transition sampling) is on
    # both.  Asynchronous: the collectives run on RCCL's stream from private copies, the next rollout does not wait for
    # them (nor for a slower rank); every handle is waited on before the clock stops.
    pending = []
    k_tr = max(0, int(getattr(args, "gather_transitions", 0)))
    tgen = torch.Generator(device=device).manual_seed(args.seed + 1000 + env_offset)
    first_obs = torch.full((B, args.n_uav, 12), -1.0, device=device)

    def gather(res, act_rows):
        gathers[0] += 1
        pending.append(uavtrack.gather_rollout_summary_async(res["ep_sums"], n_envs_total=total_envs))
        gathered_bytes[0] += res["ep_sums"].numel() * 4
        if k_tr and res.get("obs") is not None and res["reward"].shape[0] * B * args.n_uav >= k_tr:
            # the learner-side exchange: K sampled (state, action, reward, next_state) rows of this rollout
            roll = dict(obs=res["obs"], reward=res["reward"], actions=res["actions"] if "actions" in res else act_rows)
            smp = uavtrack.sample_local_transitions(first_obs, roll, k_tr, env_offset=env_offset, n_envs_total=total_envs, generator=tgen)
            h = uavtrack.gather_transitions_async(smp)
            pending.append(h)
            gathered_bytes[1] += h.nbytes_per_rank
    warm_plan, pos = launch_plan(warmup, rollout, horizon, 0)
    timed_plan, _ = launch_plan(steps, rollout, horizon, pos)
    # output buffers of every launch shape exist before the clock starts (allocation is not part of a step)
    out = {T: alloc_outputs(args, B, T, device) for T in set(timed_plan) | set(warm_plan)}
    warm = None if getattr(args, "no_device_warmup", False) else device_warmup(uavtrack, args, B, device)
    obs0 = env.reset(seed=args.seed)
    _, obs0 = run_rollouts(env, actions, warm_plan, 0, out, gather=gather, policy=args.policy, obs=obs0)
    bound = bind_plan(env, actions, timed_plan, pos, out, device) if args.policy == "given" else None
    for h in pending:
        h.wait()
    pending.clear()
    torch.cuda.synchronize(device)
    if multi:
        dist.barrier()
    g0 = gathers[0]
    gathered_bytes[0] = gathered_bytes[1] = 0
    t0 = time.perf_counter()
    launches, _ = run_rollouts(env, actions, timed_plan, pos, out, gather=gather, policy=args.policy, obs=obs0, bound=bound)
    done_h = [h.wait() for h in pending]      # the gather behind the last launch included: it is part of the region
    assert all(s.shape[0] == total_envs for s in done_h if torch.is_tensor(s))
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0       # this rank's K steps (and every gather it took part in) are done; the caller takes the MAX over ranks
    if multi:
        dist.barrier()                    # closing bracket: no rank leaves the region before the slowest has stopped its clock
    in_region = gathers[0] - g0
    assert in_region == len(gather_points(timed_plan, pos, horizon)) >= 1
    info = env.kernel_info()
    env.close()
    if warm is not None:
        warm.pop("keep")[0].close()
    gather_info = {"in_region": in_region, "collective": bool(multi),
                   "timed_region_bytes_per_rank": {"summaries": gathered_bytes[0], "transitions": gathered_bytes[1]}}
    if multi:
        # what the collectives cost on their own: blocking, back to back, outside the timed region (in it they overlap the next rollout)
        ep = torch.zeros(B, 5, device=device)
        def timed(fn, reps=5):
            fn(); torch.cuda.synchronize(device); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize(device)
            return (time.perf_counter() - t0) * 1e3 / reps
        gather_info.update({"summary_ms": timed(lambda: uavtrack.gather_rollout_summary(ep, n_envs_total=total_envs)),
                            "summary_bytes_per_rank": B * 5 * 4})
        if k_tr:
            T0 = max(set(timed_plan), key=timed_plan.count)
            roll = dict(obs=out[T0]["obs"], reward=out[T0]["reward"], actions=actions[:T0] if "actions" not in out[T0] else out[T0]["actions"])
            smp = uavtrack.sample_local_transitions(first_obs, roll, min(k_tr, T0 * B * args.n_uav), env_offset=env_offset, n_envs_total=total_envs, generator=tgen)
            gather_info["transitions_sample_ms"] = timed(lambda: uavtrack.sample_local_transitions(
                first_obs, roll, min(k_tr, T0 * B * args.n_uav), env_offset=env_offset, n_envs_total=total_envs, generator=tgen))
            gather_info["transitions_ms"] = timed(lambda: uavtrack.gather_transitions(smp))
            gather_info["transitions_bytes_per_rank"] = smp["actions"].shape[0] * 28 * 4
            gather_info["transitions_per_rank"] = int(smp["actions"].shape[0])
    return dict(wall_s=wall, launches=launches, steps=steps, geometry=info,
                timed_plan=timed_plan, warm_plan=warm_plan, device_warmup=warm, gather=gather_info)


def device_warmup(uavtrack, args, B, device, warm_ms=DEVICE_WARM_MS):
    """Untimed: `warm_ms` of back-to-back rollout launches of this workload on a throw-away handle, so that what follows
    starts at the clocks of a GPU under load, not at those of one waking up (see DEVICE_WARM_MS).  Returns a description."""
    import torch
    env = make_env(uavtrack, args, B, device)
    # (the launch length of the timed rollouts: every rollout launch of a default run is then a 200-step launch, and the
    #  rocprofv3 --kernel-trace average of the same command is the average of one launch shape, like the roofline leg's)
    T = max(1, min(getattr(args, "rollout", ROOFLINE_T), ROOFLINE_T))
    g = torch.Generator(device=device).manual_seed(1)
    actions = torch.randint(0, env.cfg.na_total, (T, B, args.n_uav), dtype=torch.int32, device=device, generator=g)
    obs = env.reset(seed=1)

    def one(out):
        env.reset(seed=1)       # (every launch an episode from the reset state, like the measured ones: under MAAC-R the
                                #  number of neighbour pairs -- the scorer's work -- depends on where the swarm is)
        if args.policy == "actor":
            return env.run_actor(T, obs, seed=1, out=out)
        if args.policy == "greedy":
            return env.run_greedy(T, seed=1)
        return env.step_many(actions, out=out)
    out = one(None)
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = one(out); e1.record()
    torch.cuda.synchronize(device)
    n = max(2, min(2000, int(warm_ms / max(e0.elapsed_time(e1), 1e-3)) + 1))
    for _ in range(n):
        out = one(out)
    # (no synchronize: the caller's launches queue up behind these; the handle is kept alive until they have run)
    return dict(ms=warm_ms, launches=n, steps_per_launch=T, keep=(env, actions, out, obs))


def roofline_leg(uavtrack, args, B, device, T=ROOFLINE_T, launches=ROOFLINE_LAUNCHES, warm_ms=DEVICE_WARM_MS):
    """`launches` rollout launches of T steps each (an episode: reset, then one launch), ALL enqueued before the first
    event is waited on and directly behind >= 3 untimed launches (at least `warm_ms` of them: the device is at its
    loaded clocks); a HIP event pair on the launch stream around every call (the resets lie outside the pairs), and,
    inside the library, around every KERNEL of the call (uavtrack_set_profiling).  Independent of --steps."""
    import torch
    env = make_env(uavtrack, args, B, device)
    g = torch.Generator(device=device).manual_seed(args.seed)
    actions = torch.randint(0, env.cfg.na_total, (T, B, args.n_uav), dtype=torch.int32, device=device, generator=g)
    out = alloc_outputs(args, B, T, device)

    def one(obs):
        if args.policy == "actor":
            return env.run_actor(T, obs, seed=42, out=out)
        if args.policy == "greedy":
            return env.run_greedy(T, seed=42)
        return env.step_many(actions, out=out)
    one(env.reset(seed=args.seed))
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    obs = env.reset(seed=args.seed)
    e0.record(); one(obs); e1.record()
    torch.cuda.synchronize(device)
    untimed = max(3, min(400, int(warm_ms / max(e0.elapsed_time(e1), 1e-3)) + 1))
    for _ in range(untimed):
        one(env.reset(seed=args.seed))
    # (MAAC-R: reading the pair counter waits for the queue -- tens of microseconds of idle, which the clocks do not notice)
    pairs0 = env.pmi_pairs_scored() if args.reward == "pmi" else 0
    env.set_profiling(True)                       # (a host-side flag)
    pairs = []
    for _ in range(launches):
        obs = env.reset(seed=args.seed)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        one(obs)
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize(device)
    ms = [e0.elapsed_time(e1) for e0, e1 in pairs]
    kernels = env.profile()
    env.set_profiling(False)
    pmi_pairs = env.pmi_pairs_scored() - pairs0 if args.reward == "pmi" else 0
    info = env.kernel_info()
    info["last_launch"] = env.launch_info()
    pmi_info = env.pmi_info() if args.reward == "pmi" else None
    env.close()
    srt = sorted(ms)
    return dict(T=T, launches=launches, untimed=untimed, ms=ms, avg_ms=sum(ms) / len(ms), median_ms=srt[len(srt) // 2],
                min_ms=srt[0], max_ms=srt[-1], pmi_pairs=pmi_pairs, geometry=info, kernels=kernels, pmi_info=pmi_info)


VERBOSE = False     # --verbose: per-launch arrays and prose notes stay on the line (the default line is kept short: the driver
                    # stores the tail of stdout, and what it must be able to read is in `configs` at the very end)


def leg_summary(roof):
    """The per-launch statistics every roofline object carries."""
    k = {name: round(v["ms"] / roof["launches"], 5) for name, v in roof["kernels"].items() if v["launches"]}
    d = {"steps_per_launch": roof["T"], "launches_timed": roof["launches"], "launches_untimed_before": roof["untimed"],
         "avg_launch_ms": roof["avg_ms"], "median_launch_ms": roof["median_ms"], "min_launch_ms": roof["min_ms"],
         "max_launch_ms": roof["max_ms"], "kernel_ms_per_launch": k}
    if VERBOSE:
        d["launch_ms"] = roof["ms"]
    return d


def pmi_roofline(args, roof, units_per_launch):
    """MAAC-R: the roofline object of the pair scorer.  `achieved` divides by the SCORER's own time (library-side HIP
    events around its launches); the whole-call figures are given beside it."""
    H = args.pmi_hidden
    flop_pair = 2.0 * (12 * H + 3 * H * H + H)               # 101 632 at H = 128
    hp = (H + 31) // 32 * 32
    # which kernel scored the pairs: the library says (uavtrack_pmi_info) -- f16 x 3 ("t3") by default, bf16 x 6 ("x6") for
    # networks beyond f16's range or when pinned, fp32 MFMA for widths the split kernels are not built for
    lib_scheme = (roof.get("pmi_info") or {}).get("scheme") or "fp32"
    scheme = {"f16x3": "t3", "bf16x6": "x6", "fp32": "fp32"}[lib_scheme]
    split = scheme != "fp32"
    call_s = sum(roof["ms"]) * 1e-3
    sc = roof["kernels"].get("scorer", {"ms": 0.0, "launches": 0})
    scorer_s = sc["ms"] * 1e-3 if sc["launches"] else call_s
    pairs = roof["pmi_pairs"]
    tf_call = pairs * flop_pair / call_s / 1e12
    tf_scorer = pairs * flop_pair / scorer_s / 1e12
    traffic, traffic_src = None, None
    try:      # HBM bytes per scorer launch from the committed PMC profile of this launch shape (profiles/r03pmi*)
        ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(
            f"{args.envs}x{args.n_uav}x{args.m_targets}_T{roof['T']}_pmi{H}_scorer")
        if ent and args.box == 2000.0 and args.dim == 2:
            traffic, traffic_src = ent["hbm_bytes_per_launch"], ent.get("source")
    except Exception:
        pass
    # what the scorer must move per pair: its 8-byte record, the two 48-byte observation rows, the 4-byte score
    alg_bytes = 108.0 * pairs / max(1, roof["launches"])
    common = {
        "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": alg_bytes,
        "physical_hbm_frac": None if traffic is None else traffic / (scorer_s / roof["launches"]) / 1e9 / HBM_PEAK_GBS,
        "flop_per_pair": flop_pair, "pairs_scored": pairs,
        "pairs_per_agent_step": pairs / (units_per_launch * roof["launches"]),
        "rescored_chunks": (roof.get("pmi_info") or {}).get("rescored_chunks", 0),
        "scorer_ms_per_launch": scorer_s * 1e3 / roof["launches"],
        "fp32_equivalent_tflops": tf_scorer, "fp32_equivalent_over_fp32_mfma_peak": tf_scorer / FP32_MFMA_PEAK_TFLOPS,
        "whole_call_fp32_equivalent_tflops": tf_call,
        **leg_summary(roof),
    }
    if split:
        # The 3H x H layer at fp32 accuracy on the 16-bit matrix cores: an fp32 operand is split into f16 (hi, lo * 2^11:
        # THREE MFMAs per fp32 product, pmi_score_h3_kernel) or bf16 (three parts: SIX, pmi_score_x6_kernel) terms.  The
        # roof that bounds it is the 16-bit matrix rate (f16 and bf16 MFMAs take the same cycles); `achieved` counts the
        # flops the matrix cores really execute, not the fp32-equivalent work.
        nprod = 6.0 if scheme == "x6" else 3.0
        # (the transposed kernel t3 also runs the 12 -> 3H branch layers on the matrix cores: 16 padded inputs x 3H units)
        executed = pairs * nprod * 2.0 * (3 * hp * hp + (16 * 3 * hp if scheme == "t3" else 0))
        return {
            "bound": "mfma", "kernel": f"pmi_score_{scheme}_kernel<{hp}>", "achieved": executed / scorer_s / 1e12,
            "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": executed / scorer_s / 1e12 / BF16_MFMA_PEAK_TFLOPS,
            "whole_call_frac": executed / call_s / 1e12 / BF16_MFMA_PEAK_TFLOPS, "mfma_products_per_fp32_product": nprod, **common,
        }
    return {
        "bound": "mfma", "kernel": f"pmi_score_kernel<{hp}>", "achieved": tf_scorer, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": tf_scorer / FP32_MFMA_PEAK_TFLOPS, "whole_call_frac": tf_call / FP32_MFMA_PEAK_TFLOPS, **common,
    }


def hbm_roofline(args, roof, B, N, M):
    """The HBM roofline object of a rollout_kernel launch shape (SURVEY 8d algorithmic bytes)."""
    bytes_unit = algorithmic_bytes_per_agent_step(N, M, args.dim)
    units = B * N * roof["T"]
    rk = roof["kernels"].get("rollout", {"ms": 0.0, "launches": 0})
    # the rollout kernel's own time (library-side events); the call-level events agree within a few microseconds
    kern_ms = rk["ms"] / rk["launches"] if rk["launches"] else roof["avg_ms"]
    achieved = bytes_unit * units / (kern_ms * 1e-3) / 1e9
    traffic, src = lookup_traffic(B, N, M, roof["T"], args)
    r = {
        "bound": "hbm", "kernel": "rollout_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved / HBM_COPY_CEILING_GBS,
        "traffic": traffic, "algorithmic_bytes_per_agent_step": bytes_unit, "agent_steps_per_launch": units,
        "algorithmic_bytes_per_launch": bytes_unit * units,
        # `frac` prices the ALGORITHMIC bytes of SURVEY 8d; the fused rollout keeps the state on chip, so the bytes that
        # really cross the HBM interface (PMC counters of this launch shape, profiles/) are fewer: this is that stream's rate
        "physical_GBs": None if traffic is None else traffic / (kern_ms * 1e-3) / 1e9,
        "physical_frac": None if traffic is None else traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "kernel_avg_ms": kern_ms,
        "frac_at_median_call": bytes_unit * units / (roof["median_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "frac_at_slowest_call": bytes_unit * units / (roof["max_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        **leg_summary(roof),
    }
    if src:
        r["traffic_source"] = src
    return r


def other_configs(uavtrack, args, device):
    """The other two single-GPU BASELINE configurations on the default line: a fixed roofline leg each for configs[2]
    (4096 x 20 x 10 MAAC-R, PMI at H = 128 = configs/MAAC-R.yaml:39 and at H = 64 = PMINet.py:21) and configs[3]
    (8192 x 50 x 25, 3-D).  ~2 s of GPU time."""
    import copy
    out = []
    for name, over in (("configs[2] MAAC-R, PMI hidden 128", dict(reward="pmi", pmi_hidden=128, envs=4096, n_uav=20, m_targets=10, dim=2)),
                       ("configs[2] MAAC-R, PMI hidden 64 (PMINetwork's default)", dict(reward="pmi", pmi_hidden=64, envs=4096, n_uav=20, m_targets=10, dim=2)),
                       ("configs[3] 3-D kinematics", dict(reward="raw", envs=8192, n_uav=50, m_targets=25, dim=3)),
                       # "also report a dense variant (box 500 m) as the PMI worst case"
                       ("configs[2] MAAC-R dense variant (500 m box), PMI hidden 128", dict(reward="pmi", pmi_hidden=128, envs=4096, n_uav=20, m_targets=10, dim=2, box=500.0))):
        a = copy.copy(args)
        a.policy, a.box = "given", 2000.0
        for k, v in over.items():
            setattr(a, k, v)
        a.cooperative = 0.0 if a.reward == "raw" else 0.3
        B, N, M = a.envs, a.n_uav, a.m_targets
        roof = roofline_leg(uavtrack, a, B, device)
        units = B * N * roof["T"]
        ent = {"config": name,
               "workload": f"{B} envs x {N} UAVs x {M} targets, {a.dim}-D, {a.box:.0f} m box, "
                           + (f"MAAC-R reciprocal (PMI H={a.pmi_hidden}) reward" if a.reward == "pmi" else "MAAC tracking reward"),
               "agent_steps_per_s": units / (roof["avg_ms"] * 1e-3), "agent_steps_per_s_at_median": units / (roof["median_ms"] * 1e-3),
               "geometry": roof["geometry"]}
        hb = hbm_roofline(a, roof, B, N, M)
        if a.reward == "pmi":
            ent["roofline"] = pmi_roofline(a, roof, units)
            ent["roofline_hbm_all_kernels"] = {k: hb[k] for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_agent_step")}
            ent["roofline_hbm_all_kernels"]["achieved"] = hb["algorithmic_bytes_per_agent_step"] * units / (roof["avg_ms"] * 1e-3) / 1e9
            ent["roofline_hbm_all_kernels"]["frac"] = ent["roofline_hbm_all_kernels"]["achieved"] / HBM_PEAK_GBS
        else:
            ent["roofline"] = hb
        out.append(ent)
    return out


def host_cores():
    """CPUs this process may really use: the cgroup quota when there is one (the GPU box
    grants 16 of the host's 256), else the affinity mask."""
    if "UAVTRACK_CPU_THREADS" in os.environ:
        return int(os.environ["UAVTRACK_CPU_THREADS"])
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(args, seconds):
    """The C oracle (oracle/uav_oracle.c: scalar fp64 port of the reference algorithm) on this
    host's cores, on a bounded sample of the same workload.  Reported, never the target."""
    import numpy as np
    from oracle import OracleConfig, OracleEnv
    # the GPU box gives one GPU's share of the host: 16 cores (os.cpu_count() reports the whole machine)
    cores = host_cores()
    N, M = args.n_uav, args.m_targets

    def run(B, T, threads):
        env = OracleEnv(OracleConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=args.cooperative, x_max=args.box, y_max=args.box,
                                     dim=args.dim, nc=3 if args.dim == 3 else 1), n_threads=threads)
        if args.reward == "pmi":
            from oracle import OraclePmi
            env.pmi = OraclePmi.from_state_dict(synthetic_pmi_state_dict(args.pmi_hidden, 42))
        env.reset_philox(seed=args.seed)
        act = np.random.RandomState(args.seed).randint(0, 12 * (3 if args.dim == 3 else 1), size=(T, B, N)).astype(np.int32)
        t0 = time.perf_counter()
        for t in range(T):
            env.step(act[t])
        return B * N * T / (time.perf_counter() - t0)

    r1 = run(64, 20, 1)                                      # calibrate
    rN = run(64 * cores, 20, cores)
    T = 200
    B = int(max(cores, min(4096, rN * seconds * 0.7 / (N * T))))
    B1 = int(max(1, min(4096, r1 * seconds * 0.3 / (N * T))))
    # whole episodes of the (capped) workload, repeated until ~20 core-seconds have been spent on this leg
    reps = int(max(1, min(16, round(20.0 / cores / (B * N * T / rN)))))
    rates = [run(B, T, cores) for _ in range(reps)]         # each: the stepping loop of one episode batch
    vN = reps / sum(1.0 / r for r in rates)                  # total agent-steps / total stepping time
    v1 = run(B1, T, 1)
    return dict(value=vN, unit="agent-steps/s", cores=cores, kind="port",
                sample=f"{reps} x ({B} envs x {N} UAVs x {M} targets x {T} steps), fp64 scalar C oracle, {cores} OpenMP threads",
                value_1core=v1, sample_1core=f"{B1} envs x {T} steps, 1 thread")


def lookup_traffic(B, N, M, T, args):
    """HBM bytes per launch from the committed PMC profile of exactly this launch shape, else None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if args.reward != "raw" or args.policy != "given" or not os.path.exists(path):
        return None, None
    try:
        tr = json.load(open(path))
    except Exception:
        return None, None
    ent = tr.get(f"{B}x{N}x{M}_T{T}" + ("_3d" if args.dim == 3 else ""))
    if not ent:
        return None, None
    return ent["hbm_bytes_per_launch"], ent.get("source", "profiles/")


def extras(uavtrack, args, B, device, bytes_unit):
    """The same kernel in other regimes (N = 1, headline workload only): one launch per step, a chip-filling
    batch, and the closed loops of SURVEY 8f-1 / 8f-2."""
    import torch
    N, M = args.n_uav, args.m_targets
    out = {}
    # one launch per env step (what a closed-loop policy would do eagerly); host-bound and noisy on a shared
    # box (25 to 80 us per step for one binary within a minute): three short runs, the best one is reported
    k = 400
    runs = [time_config(uavtrack, args, B, k, 100, 1, device) for _ in range(3)]
    r1 = min(runs, key=lambda r: r["wall_s"])
    out["per_step_launch"] = {
        "agent_steps_per_s": B * N * k / r1["wall_s"], "ms_per_step": r1["wall_s"] * 1e3 / k,
        "ms_per_step_all_runs": [r["wall_s"] * 1e3 / k for r in runs],
        "what": "uavtrack_step, one launch per step, pre-bound, best of three 400-step runs",
    }
    # a batch that fills the chip (SURVEY 8d 'bandwidth-saturating batch'), measured like the roofline leg
    # (200-step launches like the headline leg when the 17 GB of outputs fit comfortably; 50-step launches otherwise --
    # they carry the kernel's start-up cost four times as often and read ~15 % lower)
    free_b, _ = torch.cuda.mem_get_info(device)
    Bs, Ts = 65536, (200 if free_b > (64 << 30) else 50)
    rs = roofline_leg(uavtrack, args, Bs, device, T=Ts, launches=4)
    ach = bytes_unit * Bs * N * Ts / (rs["avg_ms"] * 1e-3) / 1e9
    sat_traffic, _ = lookup_traffic(Bs, N, M, Ts, args)
    out["saturating_batch"] = {
        "workload": f"{Bs} envs x {N} UAVs x {M} targets, {Ts} steps per launch",
        "agent_steps_per_s": Bs * N * Ts / (rs["avg_ms"] * 1e-3),
        "roofline_achieved_GBs": ach, "roofline_frac": ach / HBM_PEAK_GBS,
        "traffic": sat_traffic, "algorithmic_bytes_per_launch": bytes_unit * Bs * N * Ts,
        "physical_frac": None if sat_traffic is None else sat_traffic / (rs["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "avg_launch_ms": rs["avg_ms"], "geometry": rs["geometry"],
    }
    # closed loop: actor forward [B*N,12] -> sample -> uavtrack_step per step, eager
    # launches vs the same steps replayed from a HIP graph
    torch.manual_seed(args.seed)
    actor = uavtrack.ActorMLP(hidden_dim=128, action_dim=12 * (3 if args.dim == 3 else 1)).to(device)
    cl = {}
    modes = ("eager", "graph", "actor_graph", "actor_chunks") + (("greedy_graph", "greedy_chunks") if args.dim == 2 else ())
    for mode in modes:
        env = make_env(uavtrack, args, B, device)
        ro = uavtrack.BatchedRollout(env, "greedy" if mode.startswith("greedy") else actor, steps_per_graph=10,
                                     use_graph=(mode != "eager"), device_actor=mode.startswith("actor"),
                                     fuse_chunks=mode.endswith("chunks"))
        ro.reset(seed=args.seed)
        ro.run(40)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        ro.run(400)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        cl[mode] = {"agent_steps_per_s": B * N * 400 / dt, "ms_per_step": dt * 1e3 / 400}
        env.close()
    if args.dim == 2:      # the same closed loop fused into one launch per 200-step episode
        env = make_env(uavtrack, args, B, device)
        env.reset(seed=args.seed)
        env.run_greedy(200, seed=args.seed)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(5):
            env.reset(seed=args.seed)
            env.run_greedy(200, seed=args.seed)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        cl["greedy_fused"] = {"agent_steps_per_s": B * N * 1000 / dt, "ms_per_step": dt * 1e3 / 1000}
        env.close()
    # actor + environment of a whole 200-step episode in one launch (uavtrack_run_actor)
    env = make_env(uavtrack, args, B, device)
    env.set_actor(actor)
    res = env.run_actor(200, env.reset(seed=args.seed), seed=args.seed, want_terms=False)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(5):
        res = env.run_actor(200, env.reset(seed=args.seed), seed=args.seed, want_terms=False, out=res)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    cl["actor_fused"] = {"agent_steps_per_s": B * N * 1000 / dt, "ms_per_step": dt * 1e3 / 1000}
    # what bounds it: the policy's two layers are 72 v_mfma_f32_32x32x16_f16 per wavefront-step (csrc/actor.h: three f16
    # products per fp32 product, H = 128) beside the step kernel's vector work -- priced against the dense 16-bit MFMA roof in
    # executed flops, and against HBM by the algorithmic bytes of the step (the action is produced in the kernel: no 4-byte read)
    info = env.launch_info()
    waves = info["workgroups"] * (info["workgroup"] // 64)
    mfma_flops = 72 * 2 * 32 * 32 * 16 * waves * 1000
    hbm_bytes = (bytes_unit - 4) * B * N * 1000
    cl["actor_fused"]["roofline"] = {
        "bound": "mfma", "achieved": mfma_flops / dt / 1e12, "peak": BF16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": mfma_flops / dt / 1e12 / BF16_MFMA_PEAK_TFLOPS, "mfma_per_wavefront_step": 72, "wavefronts": waves,
        "hbm_frac": hbm_bytes / dt / 1e9 / HBM_PEAK_GBS,
        "note": "neither roof binds: the launch is bound by vector-instruction issue beside the MFMAs (profiles/r04actor_summary.md)"}
    env.close()
    # the paper's method end to end on the device: the same fused actor rollout under the MAAC-R reward (configs[2])
    import copy
    a2 = copy.copy(args)
    a2.reward, a2.cooperative, a2.pmi_hidden = "pmi", 0.3, 128
    env = make_env(uavtrack, a2, B, device)
    env.set_actor(actor)
    res = env.run_actor(200, env.reset(seed=args.seed), seed=args.seed, want_terms=False)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(5):
        res = env.run_actor(200, env.reset(seed=args.seed), seed=args.seed, want_terms=False, out=res)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    cl["actor_fused_maac_r"] = {"agent_steps_per_s": B * N * 1000 / dt, "ms_per_step": dt * 1e3 / 1000}
    env.close()
    if VERBOSE:
      cl["note"] = ("reference-shaped shared actor (FnnPolicyNet 12-128-12 softmax as in configs/MAAC.yaml, random "
                  "init) + categorical sample + uavtrack_step_accumulate, all on device; eager/graph = the actor "
                  "forward and the sample as PyTorch ops, graph = 10 steps per HIP-graph replay; actor_graph = the same "
                  "with the library's own actor kernel (uavtrack_actor_actions); actor_chunks = BatchedRollout(fuse_chunks=True): "
                  "every 10 steps as ONE uavtrack_run_actor launch (same bits as the 20 kernels of the graph form); actor_fused = uavtrack_run_actor, "
                  "actor and step of a whole 200-step episode in one launch (the rollout of train.operate_epoch); "
                  "greedy_graph / greedy_fused = the same two forms with the reference's C-METHOD baseline policy "
                  "(uav.py:324-369) instead of the actor")
    out["closed_loop"] = cl
    out["compat"] = compat_leg(uavtrack)
    return out


# the unmodified reference's Environment.step, timed in the build container (Intel Xeon 2.10 GHz,
# ONE core -- the reference is single-threaded Python; it cannot travel to the GPU box), ms per step
REFERENCE_STEP_MS = {"1x5x3 MAAC": 0.56, "1x20x10 MAAC": 2.40, "1x20x10 MAAC-G": 1.54, "1x20x10 MAAC-R H=128": 6.32, "1x50x25 MAAC": 5.06}


def compat_leg(uavtrack, steps=300, warm=60):
    """The loop the reference really runs (train.py:160-185): ONE environment, one `env.step(config, pmi, actions)` per step
    on the drop-in `uavtrack.Environment`, with the N `uav.get_local_state()` calls of train.py:165-166 in front of it --
    host lists in, host lists out.  Wall microseconds per step, beside the reference's own figure."""
    import random

    class Pmi:                                   # what Environment.step needs of a PMINetwork: truthiness and a state_dict
        def state_dict(self):
            return synthetic_pmi_state_dict(128, 42)
    out = []
    for name, n, m, coop, pmi in (("1x5x3 MAAC", 5, 3, 0.0, None), ("1x20x10 MAAC", 20, 10, 0.0, None), ("1x20x10 MAAC-G", 20, 10, 0.3, None),
                                  ("1x20x10 MAAC-R H=128", 20, 10, 0.3, Pmi()), ("1x50x25 MAAC", 50, 25, 0.0, None)):
        cfg = {"environment": {"n_uav": n, "m_targets": m, "x_max": 2000, "y_max": 2000, "na": 12},
               "uav": {"dt": 1, "v_max": 20, "h_max": 6, "dc": 500, "dp": 200, "alpha": 0.6, "beta": 0.2, "gamma": 0.2},
               "target": {"v_max": 5, "h_max": 6}, "cooperative": coop}
        env = uavtrack.Environment(n_uav=n, m_targets=m, x_max=2000, y_max=2000, na=12)
        random.seed(42)
        env.reset(config=cfg)
        acts = [[random.randint(0, 11) for _ in range(n)] for _ in range(warm + steps)]

        def run(k0, k1):
            for k in range(k0, k1):
                states = [u.get_local_state() for u in env.uav_list]          # train.py:165-166
                env.step(cfg, pmi, acts[k])                                   # train.py:176
            return states
        run(0, warm)
        runs = []
        for _ in range(3):                       # (a shared box: the best of three is the adapter's own cost)
            t0 = time.perf_counter()
            run(warm, warm + steps)
            runs.append((time.perf_counter() - t0) / steps * 1e6)
        us = min(runs)
        out.append({"cfg": name, "us_per_step": us, "us_per_step_all_runs": runs, "agent_steps_per_s": n / (us * 1e-6),
                    "reference_ms_per_step": REFERENCE_STEP_MS[name], "speedup_vs_reference": REFERENCE_STEP_MS[name] * 1e3 / us})
        env._env.close()
    return {"what": "uavtrack.Environment (B = 1): N get_local_state() calls + env.step(config, pmi, actions) per step, wall time; "
                    "one uavtrack_step_host call per step",
            "reference": "BASELINE.md section 2: the unmodified reference on 1 core of the build container's Xeon 2.10 GHz (no GPU)",
            "shapes": out}


def configs_summary(line, B, N, M, args):
    """Every measured configuration in a few numbers each, as the last key of the line (< 2 KB): G agent-steps/s, the
    roofline fraction of the configuration's dominant kernel and that kernel's ms per 200-step launch."""
    def r3(x):
        return None if x is None else float(f"{x:.4g}")
    rl = line["roofline"]
    out = [{"cfg": f"{B}x{N}x{M} {args.dim}D {args.reward} (timed region)", "G": r3(line["value"] / 1e9), "ms_per_step": r3(line["ms_per_step"]),
            "frac": r3(rl["frac"]), "bound": rl["bound"], "kernel_ms": r3(rl.get("kernel_avg_ms", rl.get("scorer_ms_per_launch"))),
            "roofline_leg_G": r3(B * N * rl["steps_per_launch"] / (rl["avg_launch_ms"] * 1e-3) / 1e9)}]
    if rl.get("physical_frac") is not None:
        out[0]["phys"] = r3(rl["physical_frac"])        # HBM bytes by the PMC counters / kernel time / 8 TB/s (frac: algorithmic bytes)
    for ent in line.get("other_configs") or []:
        if not isinstance(ent, dict) or "roofline" not in ent:
            continue
        r = ent["roofline"]
        e = {"cfg": ent["config"], "G": r3(ent["agent_steps_per_s"] / 1e9), "frac": r3(r["frac"]), "bound": r["bound"],
             "kernel_ms": r3(r.get("kernel_avg_ms", r.get("scorer_ms_per_launch"))), "call_ms": r3(r["avg_launch_ms"])}
        if r.get("physical_frac") is not None:
            e["phys"] = r3(r["physical_frac"])
        if "kernel_ms_per_launch" in r:
            e["kernels_ms"] = {k: r3(v) for k, v in r["kernel_ms_per_launch"].items()}
        out.append(e)
    if "saturating_batch" in line:
        sb = line["saturating_batch"]
        out.append({"cfg": "65536x20x10 2D raw (chip-filling)", "G": r3(sb["agent_steps_per_s"] / 1e9), "frac": r3(sb["roofline_frac"]),
                    "bound": "hbm", "kernel_ms": r3(sb["avg_launch_ms"])})
        if sb.get("physical_frac") is not None:
            out[-1]["phys"] = r3(sb["physical_frac"])
    if "closed_loop" in line:
        out.append({"cfg": "closed loop 4096x20x10, G agent-steps/s", **{k: r3(v["agent_steps_per_s"] / 1e9)
                                                                      for k, v in line["closed_loop"].items() if isinstance(v, dict) and "agent_steps_per_s" in v}})
    if "compat" in line:
        out.append({"cfg": "compat Environment.step B=1, us/step (reference ms/step)",
                    **{c["cfg"][2:]: [r3(c["us_per_step"]), c["reference_ms_per_step"]] for c in line["compat"]["shapes"]}})
    if "per_step_launch" in line:
        out.append({"cfg": "T=1 launches", "G": r3(line["per_step_launch"]["agent_steps_per_s"] / 1e9), "us_per_step": r3(line["per_step_launch"]["ms_per_step"] * 1e3)})
    return out


def worker(args):
    """One rank.  WORLD_SIZE / RANK / LOCAL_RANK come from the launcher (spawn_ranks or torchrun)."""
    global VERBOSE
    VERBOSE = bool(args.verbose)
    args.cooperative = 0.0 if args.reward == "raw" else 0.3       # configs/MAAC.yaml vs MAAC-G/MAAC-R.yaml
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; refusing to report a run of a different size\n")
        sys.exit(2)
    if args.selftest_launcher:
        return launcher_selftest(args, world, rank)

    import torch
    import torch.distributed as dist
    import uavtrack

    if os.environ.get("UAVTRACK_BENCH_ONE_GPU") == "1":
        local_rank = 0                      # rehearsal: every rank shares cuda:0 (gloo backend only)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the environment has no CPU path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local_rank}, {torch.cuda.device_count()} visible)")
    device = f"cuda:{local_rank}"
    torch.cuda.set_device(device)
    rccl_world = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(args.backend)
        rccl_world = dist.get_world_size()
        assert rccl_world == world

    B, N, M = args.envs, args.n_uav, args.m_targets
    res = time_config(uavtrack, args, B, args.steps, args.warmup, args.rollout, device,
                      dist=dist if world > 1 else None, env_offset=rank * B, total_envs=world * B)
    wall = torch.tensor([res["wall_s"]], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
    rank_walls = [float(wall.item())]
    if world > 1:
        allw = [torch.zeros_like(wall) for _ in range(world)]
        dist.all_gather(allw, wall)
        rank_walls = [float(w.item()) for w in allw]
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    wall_s = float(wall.item())

    if rank == 0:
        bytes_unit = algorithmic_bytes_per_agent_step(N, M, args.dim)
        value = world * B * N * args.steps / wall_s
        roof = roofline_leg(uavtrack, args, B, device)
        units_per_launch = B * N * roof["T"]
        fn = {"given": "uavtrack_step_many", "greedy": "uavtrack_run_greedy", "actor": "uavtrack_run_actor"}[args.policy]
        if args.rollout == 1 and args.policy == "given":
            fn = "uavtrack_step"
        line = {
            "metric": "env agent-steps/sec",
            "value": value,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "rccl_world_size": rccl_world,
            "backend": ("rccl" if args.backend == "nccl" else args.backend) if world > 1 else None,
            "ms_per_step_per_rank": {"min": min(rank_walls) * 1e3 / args.steps, "max": max(rank_walls) * 1e3 / args.steps},
            "gather": res["gather"],
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall_s * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{B} envs x {N} UAVs x {M} targets per GPU, {args.dim}-D, "
                            + {"raw": "MAAC tracking reward", "mean": "MAAC-G neighbour-mean reward",
                               "pmi": f"MAAC-R reciprocal (PMI H={args.pmi_hidden}) reward"}[args.reward]
                            + (f" (BASELINE configs[{4 if world > 1 else (3 if args.dim == 3 else {'raw': 1, 'mean': 1, 'pmi': 2}[args.reward])}])"),
                "envs_total": world * B,
                "box_m": args.box,
                "step": "one Environment.step of every environment of the batch",
                "launch": f"{fn}: {describe_plan(res['timed_plan'])} per call in the timed region "
                          f"({len(res['timed_plan'])} launch{'es' if len(res['timed_plan']) != 1 else ''}; reset at every episode end, "
                          f"horizon {ROOFLINE_T})"
                          + ("; MAAC-R issues three kernels per chunk (fused step, MFMA pair scorer, softmax mix)" if args.reward == "pmi" else ""),
                "timed_launch_steps": res["timed_plan"],
                "warmup_launch_steps": res["warm_plan"],
                "actions": {"given": "pre-sampled int32[T,B,N] uniform, seed 42, resident in HBM",
                            "greedy": "in-kernel C-METHOD baseline policy (uavtrack_run_greedy), closed loop",
                            "actor": f"in-kernel FnnPolicyNet 12-{args.actor_hidden}-{12 * (3 if args.dim == 3 else 1)} actor, random init, "
                                     "Categorical sample (uavtrack_run_actor), closed loop"}[args.policy],
                "outputs": "obs[T,B,N,12] reward[T,B,N] terms[T,3,B,N] covered[T,B] done[T,B] ep_sums[B,5], all written",
                "parallelism": (f"env-sharded x{world}, {'RCCL' if args.backend == 'nccl' else args.backend} "
                                f"all-gather of ep_sums behind every rollout -- {res['gather']['in_region']} inside the timed region, the one "
                                f"behind its last launch included and waited for before the clock stops") if world > 1 else "1 GPU",
                "geometry": res["geometry"],
                "device_warmup": ("none (--no-device-warmup): the timed region starts from the GPU's idle power state" if res["device_warmup"] is None else
                                  f"{res['device_warmup']['launches']} untimed {res['device_warmup']['steps_per_launch']}-step launches of this workload on a "
                                  f"separate handle (~{res['device_warmup']['ms']:.0f} ms) directly ahead of the {args.warmup} warm-up steps: an MI355X "
                                  "that has idled runs its next ~20 ms of launches 10-20 % slow (tools/drift.py)"),
            },
            "roofline": hbm_roofline(args, roof, B, N, M),
        }
        if args.reward == "pmi":
            line["roofline_hbm_all_kernels"] = line["roofline"]
            line["roofline"] = pmi_roofline(args, roof, units_per_launch)
        if args.policy == "actor":
            # the actor adds 2*(12*H + H*16*tiles) fp32 MFMA flops per agent-step (actions padded to 16-row tiles: 1 in 2-D, 3 in 3-D)
            Hp = (args.actor_hidden + 15) // 16 * 16
            rl = line["roofline"]
            rl["actor_mfma_flop_per_agent_step"] = 2 * (12 * Hp + Hp * 16 * (3 if args.dim == 3 else 1))
            rl["actor_mfma_tflops"] = rl["actor_mfma_flop_per_agent_step"] * units_per_launch / (roof["avg_ms"] * 1e-3) / 1e12
            rl["actor_mfma_frac_of_fp32_mfma_peak"] = rl["actor_mfma_tflops"] / FP32_MFMA_PEAK_TFLOPS
            rl["note"] = ("closed-loop launch: environment step (HBM roofline above) plus the policy network on the "
                          "fp32 matrix cores (peak 157.3 TFLOP/s); both figures are over the whole launch")
        default_workload = (world == 1 and args.reward == "raw" and args.policy == "given" and args.dim == 2 and args.box == 2000.0
                            and (B, N, M) == (4096, 20, 10))
        # (the optional legs must never cost the run its line: a failure is reported in place)
        def guarded(key, fn):
            try:
                return fn()
            except Exception as exc:                      # noqa: BLE001 -- reported, not swallowed
                import traceback
                sys.stderr.write(f"bench.py: optional leg {key!r} failed:\n{traceback.format_exc()}")
                return {"error": f"{type(exc).__name__}: {exc}"}
        if default_workload and not args.no_other_configs:
            line["other_configs"] = guarded("other_configs", lambda: other_configs(uavtrack, args, device))
        if world == 1 and not args.no_extras and args.reward != "pmi" and args.policy == "given":
            ex = guarded("extras", lambda: extras(uavtrack, args, B, device, bytes_unit))
            line.update(ex if "error" not in ex else {"extras_error": ex["error"]})
        if world == 1 and not args.no_cpu_baseline:
            cb = guarded("cpu_baseline", lambda: cpu_baseline(args, args.cpu_seconds))
            line["cpu_baseline"] = cb
            if "value" in cb:
                cb["gpu_over_cpu"] = value / cb["value"]
        line["configs"] = configs_summary(line, B, N, M, args)       # LAST key: the tail of stdout is what the driver's record keeps
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args, argv))        # the parent makes no GPU call, before or after
    worker(args)


if __name__ == "__main__":
    main()
