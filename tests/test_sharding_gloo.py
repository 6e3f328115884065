"""world_size-2 gloo rehearsal of the multi-GPU path on CPU: rank r owns a contiguous
range of global env ids, resets it from the shared seed, and the end-of-rollout gather
reassembles per-env summaries in global order.  The per-rank compute stands in for the
HIP rollout with the oracle (tests may do that; the product never does)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "marl-uavs-targets-tracking_amd")]
    from oracle import OracleConfig, OracleEnv
    from uavtrack.sharding import gather_rollout_summary, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = shard_range(total, rank, world)
    env = OracleEnv(OracleConfig(n_envs=cnt, n_uav=5, m_targets=3))
    env.reset_philox(seed=42, env_offset=off)
    acts = np.random.RandomState(1).randint(0, 12, size=(4, total, 5))[:, off:off + cnt]
    ep = np.zeros((cnt, 5))
    for t in range(4):
        out = env.step(acts[t])
        ep[:, 0] += out["reward"].mean(1)
        ep[:, 1:4] += out["terms"].mean(2).T
        ep[:, 4] += out["covered"]
    # asynchronous form: start the exchange, overwrite the source (as the next rollout would), then wait
    from uavtrack.sharding import gather_rollout_summary_async
    src = torch.from_numpy(ep.copy())
    handle = gather_rollout_summary_async(src, n_envs_total=total)
    src.zero_()
    full = gather_rollout_summary(torch.from_numpy(ep), n_envs_total=total)
    assert torch.equal(handle.wait(), full) and handle.wait() is handle.wait()
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_gather_matches_unsharded(total):
    from oracle import OracleConfig, OracleEnv
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    env = OracleEnv(OracleConfig(n_envs=total, n_uav=5, m_targets=3))
    env.reset_philox(seed=42)
    acts = np.random.RandomState(1).randint(0, 12, size=(4, total, 5))
    ep = np.zeros((total, 5))
    for t in range(4):
        out = env.step(acts[t])
        ep[:, 0] += out["reward"].mean(1)
        ep[:, 1:4] += out["terms"].mean(2).T
        ep[:, 4] += out["covered"]
    np.testing.assert_array_equal(got, ep)
