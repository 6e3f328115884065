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


def _transition_worker(rank, world, port, total, k, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "marl-uavs-targets-tracking_amd")]
    from uavtrack.sharding import gather_transitions, gather_transitions_async, sample_local_transitions, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    off, cnt = shard_range(total, rank, world)
    full = _fake_rollout(total)                                        # every rank can build the unsharded rollout (seeded)
    mine = {key: v[:, off:off + cnt].contiguous() for key, v in full["out"].items()}
    sample = sample_local_transitions(full["obs_in"][off:off + cnt], mine, k, env_offset=off, n_envs_total=total,
                                      generator=torch.Generator().manual_seed(100 + rank))
    h = gather_transitions_async(sample)
    got = gather_transitions(sample)
    assert all(torch.equal(h.wait()[key], got[key]) for key in got) and h.nbytes_per_rank == k * 28 * 4
    if rank == 0:
        q.put({key: v.numpy() for key, v in got.items()})
    dist.barrier()
    dist.destroy_process_group()


def _fake_rollout(total, T=6, N=5):
    g = torch.Generator().manual_seed(3)
    out = dict(obs=torch.randn(T, total, N, 12, generator=g), actions=torch.randint(0, 12, (T, total, N), dtype=torch.int32, generator=g),
               reward=torch.randn(T, total, N, generator=g))
    return dict(out=out, obs_in=torch.randn(total, N, 12, generator=g))


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_transition_gather(total):
    """The learner-side exchange of a sharded rollout (uavtrack.sharding.gather_transitions): each rank samples K of its own
    (state, action, reward, next_state) transitions (train.py:176-180), one all-gather hands every rank all 2 K rows; every
    gathered row is the row of the UNSHARDED rollout its global index names, ranks sample only their own environments, no
    transition twice.  Even and uneven shards."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "marl-uavs-targets-tracking_amd")]
    from uavtrack.replay import DeviceReplayBuffer, transitions_from_rollout
    from uavtrack.sharding import shard_range
    k, N = 40, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_transition_worker, args=(r, 2, port, total, k, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = _fake_rollout(total)
    tr = transitions_from_rollout(full["obs_in"], full["out"])             # flat [T * total * N] in (t, b, i) order
    idx = torch.from_numpy(got["index"])
    assert idx.shape == (2 * k,) and idx.unique().numel() == 2 * k
    for key, name in (("states", "states"), ("actions", "actions"), ("rewards", "rewards"), ("next_states", "next_states")):
        assert torch.equal(torch.from_numpy(got[key]), tr[name][idx]), key
    b = (idx // N) % total
    for r in range(2):
        off, cnt = shard_range(total, r, 2)
        assert ((b[r * k:(r + 1) * k] >= off) & (b[r * k:(r + 1) * k] < off + cnt)).all()
    buf = DeviceReplayBuffer(1000, "cpu")
    buf.add({key: torch.from_numpy(got[key]) for key in ("states", "actions", "rewards", "next_states")})
    assert buf.size() == 2 * k
