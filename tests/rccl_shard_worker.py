#!/usr/bin/env python3
"""One rank of the multi-GPU shard test (tests/test_hip_round3.py): steps its shard of a batch on its own GPU and takes
part in the end-of-rollout gather (uavtrack.sharding; RCCL when --backend nccl).  Rank 0 writes the gathered
[B_total, 5] episode sums to --out.  Started as a child process with the torchrun environment contract
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) by a parent that does not hand its GPU context over.

    rccl_shard_worker.py --backend nccl|gloo --envs B_total --steps T --out file.npy [--one-gpu]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--envs", type=int, required=True)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--out", required=True)
    ap.add_argument("--one-gpu", action="store_true", help="every rank shares cuda:0 (gloo rehearsal on a 1-GPU box)")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if a.one_gpu else int(os.environ.get("LOCAL_RANK", rank))

    import numpy as np
    import torch
    import torch.distributed as dist
    import uavtrack
    from uavtrack.sharding import gather_rollout_summary, gather_rollout_summary_async, shard_range

    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    if a.backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(a.backend)
    assert dist.get_world_size() == world

    off, cnt = shard_range(a.envs, rank, world)
    cfg = uavtrack.EnvConfig(n_envs=cnt, n_uav=20, m_targets=10, cooperative=0.3, env_offset=off)
    env = uavtrack.BatchedUavEnv(cfg, str(device))
    env.reset(seed=42)
    # the same global action tensor on every rank (seeded), sliced to the shard
    g = torch.Generator(device="cpu").manual_seed(7)
    act = torch.randint(0, 12, (a.steps, a.envs, 20), dtype=torch.int32, generator=g)[:, off:off + cnt].contiguous().to(device)
    out = env.step_many(act)
    # asynchronous form first (the source is overwritten at once, as the next rollout would), then the blocking one
    src = out["ep_sums"].clone()
    handle = gather_rollout_summary_async(src, n_envs_total=a.envs)
    src.zero_()
    full = gather_rollout_summary(out["ep_sums"], n_envs_total=a.envs)
    assert torch.equal(handle.wait(), full)
    # the learner-side exchange: K sampled transitions per rank, one all-gather (uavtrack.sharding.gather_transitions)
    from uavtrack.sharding import gather_transitions, sample_local_transitions
    obs_in = torch.full((cnt, 20, 12), -1.0, device=device)            # (any fixed first observation will do for the check)
    roll = dict(obs=out["obs"], actions=act, reward=out["reward"])
    sample = sample_local_transitions(obs_in, roll, 512, env_offset=off, n_envs_total=a.envs,
                                      generator=torch.Generator(device=device).manual_seed(50 + rank))
    tr = gather_transitions(sample)
    assert tr["states"].shape == (world * 512, 12) and tr["states"].device == device
    assert torch.equal(tr["next_states"][rank * 512:(rank + 1) * 512], sample["next_states"])
    torch.cuda.synchronize(device)
    if rank == 0:
        np.save(a.out, full.cpu().numpy())
        np.savez(a.out + ".transitions.npz", **{k: v.cpu().numpy() for k, v in tr.items()})
    dist.barrier()
    dist.destroy_process_group()
    env.close()


if __name__ == "__main__":
    main()
