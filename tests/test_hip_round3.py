"""GPU parity, round-3 additions (all through the C ABI, like tests/test_hip_parity.py):

* the reference-shaped adapter driven with a PMINetwork-shaped MODULE, i.e. the MAAC-R call shape of train.py:176
  (`env.step(config, pmi, action_list)`), against the reference's recorded rewards (g4);
* every per-step constant away from configs/*.yaml, and the clip's N / M taken from a config that differs from the
  environment's own sizes (environment.py:207-220): against the reference's recorded outputs (g8) and the oracle;
* the scorer alone (uavtrack_pmi_inference = PMINetwork.inference, PMINet.py:64-72) on adversarial weights -- large
  cancelling 3H-term sums, magnitudes over 2^-20 .. 2^4 -- against an fp64 forward;
* the checkpoint surface: episode counters of the automatic reset, target-trace buffer lifetime;
* BASELINE configs[4]'s batch (32 768 x 20 x 10) on one GPU as 8 sequential 4 096-env shards == one 32 768-env handle,
  bitwise; the two-rank shard + gather path as child processes (RCCL when two GPUs are visible, gloo sharing the one
  GPU otherwise).
"""
import gc
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import adversarial_pmi_state_dict, load_golden, pmi_forward_fp64
from oracle import OracleConfig, OracleEnv, OraclePmi
from test_hip_parity import ATOL, Tally, compare_step, host, inject
from test_oracle_golden import oracle_cfg_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def uavtrack():
    import uavtrack
    return uavtrack


def test_compat_environment_maac_r_with_pmi_module(uavtrack, pmi_state_dict):
    """train.py:176 under MAAC-R: env.step(config, pmi, action_list) with `pmi` a PMINetwork-shaped torch module
    (uavtrack.make_pmi_net, state_dict-compatible with PMINet.py:20-38) holding the reference's recorded weights, over
    g4's recorded states and actions; rewards against g4's recorded rewards."""
    import random
    z, meta = load_golden("g4_n20m10_pmi")
    N, M = meta["n_uav"], meta["m_targets"]
    ref_cfg = meta["cfg"]
    assert ref_cfg["cooperative"] == 0.3
    pmi = uavtrack.make_pmi_net(128)
    missing = pmi.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in pmi_state_dict.items()}, strict=False)
    assert all("num_batches_tracked" in k for k in missing.missing_keys) and not missing.unexpected_keys
    pmi.eval()
    env = uavtrack.Environment(n_uav=N, m_targets=M, x_max=2000, y_max=2000, na=12)
    random.seed(42)
    assert env.reset(config=ref_cfg) is None
    orc = OracleEnv(OracleConfig(n_envs=1, n_uav=N, m_targets=M, cooperative=0.3))
    compared = moved = 0
    # reset() saw no network (MAAC-G handle); the first step with `pmi` switches the adapter to MAAC-R (the mode follows
    # the argument's truthiness, uav.py:319) and carries the state over -- a throw-away step does that here
    env.step(ref_cfg, pmi, [int(a) for a in z["actions"][0, 0]])
    for e in (0, 3):
        for t in range(12):
            st = {k: z[k][e, t][None] for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")}
            env._env.set_state(**st)
            act = [int(a) for a in z["actions"][e, t]]
            nxt, reward, covered = env.step(ref_cfg, pmi, act)
            assert env._cfg.resolved_mode() == uavtrack.RewardMode.PMI
            assert isinstance(covered, int) and len(nxt) == N and nxt[0].shape == (12,)
            assert set(reward) == {"rewards", "target_tracking_reward", "boundary_punishment", "duplicate_tracking_punishment"}
            inject(orc, {k: np.asarray(v, dtype=np.float32) for k, v in st.items()})
            if orc.step(np.asarray(act, dtype=np.int32)[None])["margin"][0] <= 5e-3:
                continue                                   # an fp32 knife edge in this step: not comparable
            np.testing.assert_allclose(reward["rewards"], z["reward"][e, t], rtol=0, atol=2e-5, err_msg=f"e{e} t{t}")
            np.testing.assert_allclose(np.array(nxt), z["obs"][e, t], rtol=0, atol=2e-5)
            for k, row in zip(("target_tracking_reward", "boundary_punishment", "duplicate_tracking_punishment"), z["terms"][e, t]):
                np.testing.assert_allclose(reward[k], row, rtol=0, atol=2e-5)
            assert covered == int(z["covered"][e, t])
            compared += 1
            moved += int(np.abs(z["reward"][e, t] - z["raw"][e, t]).max() > 1e-3)
    assert compared >= 16 and moved >= 4          # and the PMI term really shaped the rewards that were compared
    # an episode later the learner has updated the network (train.py:262): new weights must reach the device
    with torch.no_grad():
        pmi.fc2.weight.mul_(-3.0)
    env.reset(config=ref_cfg)
    env._env.set_state(**{k: z[k][0, 5][None] for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")})
    _, r2, _ = env.step(ref_cfg, pmi, [int(a) for a in z["actions"][0, 5]])
    assert np.abs(np.array(r2["rewards"]) - z["reward"][0, 5]).max() > 1e-4


def _env_kw(case, n_envs):
    kw = oracle_cfg_of(case, n_envs)
    return kw


@pytest.mark.parametrize("idx", [0, 1, 2])
def test_nondefault_constants_vs_reference_and_oracle(uavtrack, pmi_state_dict_h64, idx):
    """g8 (dt .5, v_max 13, h_max pi/5, dp 173.3, dc 411.7, alpha/beta/gamma .5/.3/.2, target v_max 7, na 9, 1500 x 1100 box,
    clip N / M from a config that differs from the environment's sizes -- environment.py:207-220, configs/MAAC-R.yaml:9-28):
    every recorded (episode, step) as one batch against the reference's outputs, then free-running teacher-forced steps
    against the oracle with the same constants."""
    z, meta = load_golden("g8_nondefault")
    case = meta["cases"][idx]
    pre = case["name"] + "__"
    N, M, T, E = case["n_uav"], case["m_targets"], case["steps"], len(case["seeds"])
    B = E * T
    kw = _env_kw(case, B)
    mode = uavtrack.RewardMode.PMI if case["pmi"] else None
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(reward_mode=mode, **kw))
    orc = OracleEnv(OracleConfig(**kw), n_threads=8)
    if case["pmi"]:
        env.set_pmi(pmi_state_dict_h64)
        orc.pmi = OraclePmi.from_state_dict(pmi_state_dict_h64)
    pick = lambda k: z[pre + k][:, :T].reshape(B, -1)
    env.set_state(**{k: pick(k) for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")})
    act = z[pre + "actions"].reshape(B, N).astype(np.int32)
    inject(orc, host(env.get_state()))
    ref = orc.step(act)
    ok = ref["margin"] > 5e-3
    assert ok.mean() > 0.9
    obs, rew, _ = env.step(torch.from_numpy(act))
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    terms, cov = env.info["terms"].cpu().numpy(), env.info["covered"].cpu().numpy()
    # north_star's 1e-5 against the oracle restarted from the same fp32 state; 2e-5 against the recorded fp64-state outputs
    np.testing.assert_allclose(rew[ok], ref["reward"][ok], rtol=0, atol=ATOL)
    np.testing.assert_allclose(obs[ok], ref["obs"][ok], rtol=0, atol=ATOL)
    np.testing.assert_allclose(rew[ok], z[pre + "reward"].reshape(B, N)[ok], rtol=0, atol=2e-5)
    np.testing.assert_allclose(obs[ok], z[pre + "obs"].reshape(B, N, 12)[ok], rtol=0, atol=2e-5)
    np.testing.assert_allclose(terms[:, ok], z[pre + "terms"].transpose(2, 0, 1, 3).reshape(3, B, N)[:, ok], rtol=0, atol=2e-5)
    np.testing.assert_array_equal(cov[ok], z[pre + "covered"].reshape(B)[ok])
    st = host(env.get_state())
    nxt = lambda k: z[pre + k][:, 1:T + 1].reshape(B, -1)
    np.testing.assert_allclose(st["ux"], nxt("ux"), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(st["ty"], nxt("ty"), rtol=1e-5, atol=1e-4)
    np.testing.assert_array_equal(st["ua"], nxt("ua"))
    # free-running from a device reset, teacher-forced against the oracle (turn rates of na = 9, the half time step ...)
    env.reset(seed=77)
    rng = np.random.RandomState(5)
    tally = Tally()
    for t in range(5):
        a = rng.randint(0, kw["na"], size=(B, N)).astype(np.int32)
        compare_step(env, orc, a, f"g8 {case['name']} t{t}", tally=tally)
    tally.check(f"g8 {case['name']}")
    env.close()


@pytest.mark.parametrize("hidden", [128, 64, 96, 200])
def test_pmi_scorer_adversarial_cancellation(uavtrack, hidden):
    """The scorer alone (uavtrack_pmi_inference) on weights built against the bf16 x 6 split: fc1 magnitudes over
    2^-20 .. 2^4, every 3H-term sum a ~1000-fold cancellation (conftest.adversarial_pmi_state_dict).  No fp32 evaluation
    can hold an ABSOLUTE 1e-5 there (one rounding of a partial sum is already 2^-24 of its size); what is asserted is
    1e-5 RELATIVE to sum |terms| -- and, on the same weights scaled down to O(1) sums, the plain 1e-5 of north_star.
    H = 200 runs on the fp32-MFMA kernel (same bound), the others on the bf16 x 6 kernel."""
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=1, n_uav=2, m_targets=1, cooperative=0.3,
                                                    reward_mode=uavtrack.RewardMode.PMI))
    rng = np.random.RandomState(hidden)
    n = 5000 + hidden                                   # (not a multiple of the 32-pair tile)
    x = rng.uniform(-1.0, 1.0, (n, 12)).astype(np.float32)
    x[: n // 4] *= rng.uniform(0.0, 4.5, (n // 4, 1)).astype(np.float32) ** 2      # products la_i * la_j reach ~ 20 (x / dc squared)
    xd = torch.from_numpy(x).cuda()
    worst = {}
    for label, scale_w in (("adversarial", 1.0), ("well_scaled", 2.0 ** -10)):
        sd = adversarial_pmi_state_dict(hidden, seed=hidden)
        sd["fc1.weight"] = (sd["fc1.weight"] * np.float32(scale_w)).astype(np.float32)
        env.set_pmi(sd)
        got = env.pmi_inference(xd).cpu().numpy().astype(np.float64)
        ref, mag = pmi_forward_fp64(sd, x, want_scale=True)
        err = np.abs(got - ref)
        worst[label] = (err.max(), (err / np.maximum(mag, 1.0)).max(), mag.max(), np.abs(ref).max())
        assert np.isfinite(got).all()
        assert (err <= 1e-5 * np.maximum(mag, 1.0)).all(), (label, worst[label])
        if label == "well_scaled":
            assert mag.max() < 50.0 and err.max() <= 1e-5 * max(1.0, mag.max() / 4.0), worst[label]
    if os.environ.get("UAVTRACK_TEST_REPORT"):
        print(f"[x6 adversarial] H={hidden}: " + "; ".join(
            f"{k}: max err {v[0]:.2e}, max err / sum|terms| {v[1]:.2e}, sum|terms| <= {v[2]:.0f}, |score| <= {v[3]:.2f}" for k, v in worst.items()))
    # the same entry point against the MAAC-R step itself: scores feed the softmax of uav.py:287, so a step with these
    # weights must still agree with the oracle's reward at 1e-5 (rewards are convex combinations: bounded sensitivity)
    env.close()
    kw = dict(n_envs=64, n_uav=20, m_targets=10, cooperative=0.3, x_max=600.0, y_max=600.0)
    sd = adversarial_pmi_state_dict(hidden, seed=hidden)
    sd["fc1.weight"] = (sd["fc1.weight"] * np.float32(2.0 ** -10)).astype(np.float32)
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(reward_mode=uavtrack.RewardMode.PMI, **kw))
    env.set_pmi(sd)
    env.reset(seed=3)
    orc = OracleEnv(OracleConfig(**kw), n_threads=8)
    orc.pmi = OraclePmi.from_state_dict(sd)
    for t in range(3):
        compare_step(env, orc, rng.randint(0, 12, size=(64, 20)).astype(np.int32), f"adversarial PMI H{hidden} t{t}")
    env.close()


@pytest.mark.parametrize("hidden", [128, 64])
def test_pmi_scorer_block_scales_at_their_extremes(uavtrack, hidden, pmi_state_dict, pmi_state_dict_h64):
    """pmi_score_t3_kernel folds powers of two into its f16 planes (S1 into the branch layers from the activation bound, T
    into fc1 from max |w|: uavtrack_set_pmi_weights) so that operand remainders need no scaling on the device.  The
    reference-initialised network with its layers scaled far up and far down -- S1 and T at both ends of their ranges,
    activations from ~1e-4 to ~1e4 of the nominal ones, inputs up to the (x / dc)^2 ~ 250 of a UAV far outside the box --
    against fp64: 1e-5 relative to sum |terms| (floor 1: the plain 1e-5 of north_star where the sums are O(1))."""
    base = pmi_state_dict if hidden == 128 else pmi_state_dict_h64
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=1, n_uav=2, m_targets=1, cooperative=0.3,
                                                    reward_mode=uavtrack.RewardMode.PMI))
    rng = np.random.RandomState(7 + hidden)
    n = 4096 + 17
    x = rng.uniform(-1.0, 1.0, (n, 12)).astype(np.float32)
    x[: n // 8, 9:11] *= rng.uniform(0.0, 16.0, (n // 8, 1)).astype(np.float32) ** 2        # boundary-state products of a far-away UAV
    x[n // 8: n // 4] *= np.float32(1e-3)                                                    # ... and inputs that are all tiny
    xd = torch.from_numpy(x).cuda()
    report = []
    for label, s_branch, s_fc1 in (("nominal", 1.0, 1.0), ("branches x 2^5", 32.0, 1.0), ("branches x 2^-12", 2.0 ** -12, 1.0),
                                   ("fc1 x 2^9", 1.0, 512.0), ("fc1 x 2^-14", 1.0, 2.0 ** -14), ("both up", 16.0, 64.0)):
        sd = {k: np.array(v, dtype=np.float32) for k, v in base.items()}
        for bn in ("bn_comm", "bn_obs", "bn_boundary_state"):               # the folded branch layer = s_branch x the nominal one
            sd[bn + ".weight"] *= np.float32(s_branch)
            sd[bn + ".bias"] *= np.float32(s_branch)
        sd["fc1.weight"] *= np.float32(s_fc1)
        env.set_pmi(sd)
        got = env.pmi_inference(xd).cpu().numpy().astype(np.float64)
        ref, mag = pmi_forward_fp64(sd, x, want_scale=True)
        err = np.abs(got - ref)
        assert np.isfinite(got).all(), label
        report.append((label, err.max(), (err / np.maximum(mag, 1.0)).max(), mag.max()))
        assert (err <= 1e-5 * np.maximum(mag, 1.0)).all(), report[-1]
    if os.environ.get("UAVTRACK_TEST_REPORT"):
        print(f"[t3 block scales] H={hidden}: " + "; ".join(f"{l}: max err {e:.2e}, / sum|terms| {r:.2e} (sum|terms| <= {m:.3g})" for l, e, r, m in report))
    env.close()


@pytest.mark.parametrize("hidden", [128, 64])
def test_pmi_inference_tiny_batches(uavtrack, hidden, pmi_state_dict, pmi_state_dict_h64):
    """The scorer's tile loop at its edges: fewer pairs than one 32-pair tile, exactly one, one more, fewer tiles than
    workgroups (most workgroups then run the prologue only), and the empty batch."""
    sd = pmi_state_dict if hidden == 128 else pmi_state_dict_h64
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=1, n_uav=2, m_targets=1, cooperative=0.3,
                                                    reward_mode=uavtrack.RewardMode.PMI))
    env.set_pmi(sd)
    rng = np.random.RandomState(hidden)
    for n in (1, 2, 31, 32, 33, 64, 65, 257, 8191):
        x = rng.uniform(-1.0, 1.0, (n, 12)).astype(np.float32)
        got = env.pmi_inference(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
        ref = pmi_forward_fp64(sd, x)
        assert got.shape == (n,) and np.abs(got - ref).max() < 1e-5, (n, np.abs(got - ref).max())
    assert env.pmi_inference(torch.empty(0, 12, device="cuda")).shape == (0,)
    env.close()


def test_checkpoint_restores_autoreset_episode_counters(uavtrack):
    """get_state() / set_state() carry the per-environment episode numbers that key the Philox counter of the automatic
    reset (uavtrack_get_episodes / uavtrack_set_episodes): a FRESH handle restored from a checkpoint taken mid-run
    continues -- through further automatic resets -- exactly like the run the checkpoint was taken from."""
    B, N, M, H = 37, 20, 10, 6
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3, horizon=H)
    a = uavtrack.BatchedUavEnv(cfg)
    a.reset(seed=5, episode=11)
    # stagger the episodes so that environments turn over at different steps
    st = a.get_state()
    st["step_count"] = torch.arange(B, dtype=torch.int32, device="cuda") % H
    a.set_state(**st)
    g = torch.Generator("cuda").manual_seed(1)
    act1 = torch.randint(0, 12, (2 * H + 1, B, N), dtype=torch.int32, device="cuda", generator=g)
    act2 = torch.randint(0, 12, (2 * H + 3, B, N), dtype=torch.int32, device="cuda", generator=g)
    a.step_many(act1, auto_reset_seed=99)
    ckpt = {k: v.clone() for k, v in a.get_state().items()}
    assert int(ckpt["episode"].min()) >= 12 and int(ckpt["episode"].max()) >= 13      # everyone reset at least once since
    want = a.step_many(act2, auto_reset_seed=99)
    b = uavtrack.BatchedUavEnv(cfg)                       # fresh handle: episode counters start at 0
    b.set_state(**ckpt)
    got = b.step_many(act2, auto_reset_seed=99)
    for k in ("obs", "reward", "terms", "covered", "done", "ep_sums"):
        assert torch.equal(got[k], want[k]), k
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    # without the counters the restored run diverges at its first automatic reset (what ADVICE r2 pointed out)
    c = uavtrack.BatchedUavEnv(cfg)
    c.set_state(**{k: v for k, v in ckpt.items() if k != "episode"})
    assert not torch.equal(c.step_many(act2, auto_reset_seed=99)["obs"], want["obs"])
    # and the host-side counter moved past the restored numbers: a plain reset() does not replay a used episode
    assert b._episode > int(ckpt["episode"].max())


def test_target_trace_buffer_lifetime_and_restore(uavtrack):
    """The library keeps the RAW pointer of the installed target trace: the Python object must keep the tensor alive, and a
    temporary trace (want_targets=True) must give an installed one back afterwards."""
    B, N, M = 16, 5, 3
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M))
    env.reset(seed=2)
    env.set_target_trace(torch.full((4, B, M, 2), -7.0, device="cuda"))      # the caller drops its reference at once
    gc.collect()
    junk = [torch.full((4, B, M, 2), 123.0, device="cuda") for _ in range(8)]   # would reuse the block had it been freed
    act = torch.randint(0, 12, (3, B, N), dtype=torch.int32, device="cuda")
    out = env.step_many(act, want_targets=True)                               # temporary trace ...
    assert env._trace is not None and env._trace.shape == (4, B, M, 2)        # ... and the installed one is back
    assert all(bool((j == 123.0).all()) for j in junk)
    env.step(act[0])
    st = env.get_state()
    assert torch.equal(env._trace[0, :, :, 0], st["tx"]) and torch.equal(env._trace[0, :, :, 1], st["ty"])
    assert bool((env._trace[1:] == -7.0).all())
    assert out["targets"].shape == (3, B, M, 2) and bool((out["targets"] != -7.0).all())
    env.set_target_trace(None)
    assert env._trace is None
    env.close()


def test_configs4_batch_as_eight_sequential_shards(uavtrack):
    """BASELINE configs[4]'s workload -- 32 768 envs x 20 UAVs x 10 targets, 4 096 per GPU -- on ONE GPU: eight
    4 096-env shards stepped one after the other reproduce the single 32 768-env handle bit for bit (outputs, episode
    sums, final state), which is the whole multi-GPU contract apart from the gather."""
    from uavtrack.sharding import shard_range
    Btot, N, M, T, R = 32768, 20, 10, 10, 8
    cfg = uavtrack.EnvConfig(n_envs=Btot, n_uav=N, m_targets=M, cooperative=0.3)
    full = uavtrack.BatchedUavEnv(cfg)
    full.reset(seed=42)
    act = torch.randint(0, 12, (T, Btot, N), dtype=torch.int32, device="cuda", generator=torch.Generator("cuda").manual_seed(4))
    out = full.step_many(act)
    fs = full.get_state()
    assert torch.isfinite(out["obs"]).all() and float(out["reward"].abs().max()) <= 1.0
    for r in range(R):
        off, cnt = shard_range(Btot, r, R)
        assert cnt == 4096
        sh = uavtrack.BatchedUavEnv(cfg.with_(n_envs=cnt, env_offset=off))
        sh.reset(seed=42)
        o = sh.step_many(act[:, off:off + cnt].contiguous())
        for k in ("obs", "reward", "covered", "done"):
            assert torch.equal(o[k], out[k][:, off:off + cnt]), (r, k)
        assert torch.equal(o["terms"], out["terms"][:, :, off:off + cnt]), r
        assert torch.equal(o["ep_sums"], out["ep_sums"][off:off + cnt]), r
        ss = sh.get_state()
        for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th", "step_count"):
            assert torch.equal(ss[k], fs[k][off:off + cnt]), (r, k)
        sh.close()
    full.close()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two_ranks(backend, one_gpu, total, steps, out_path):
    port = _free_port()
    procs = []
    for r in range(2):
        envv = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", LOCAL_WORLD_SIZE="2",
                    MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        envv.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, os.path.join(ROOT, "tests", "rccl_shard_worker.py"), "--backend", backend, "--envs", str(total),
               "--steps", str(steps), "--out", out_path] + (["--one-gpu"] if one_gpu else [])
        procs.append(subprocess.Popen(cmd, env=envv, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        logs.append(o.decode("utf-8", "replace")[-2000:])
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(logs)
    return np.load(out_path)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_shard_and_gather(uavtrack, tmp_path, backend):
    """Two ranks as child processes, each stepping its shard of an uneven batch on its own handle, then
    gather_rollout_summary (blocking and asynchronous): the gathered [B_total, 5] equals the unsharded handle's episode
    sums bit for bit.  `nccl` = RCCL over xGMI, needs two visible GPUs (skipped on the one-GPU box); `gloo` runs the very
    same worker with both ranks sharing the one GPU, so everything but the transport is exercised wherever this suite runs."""
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL leg needs two GPUs")
    total, steps = 8191, 8
    got = _run_two_ranks(backend, backend == "gloo", total, steps, str(tmp_path / f"gather_{backend}.npy"))
    cfg = uavtrack.EnvConfig(n_envs=total, n_uav=20, m_targets=10, cooperative=0.3)
    env = uavtrack.BatchedUavEnv(cfg)
    env.reset(seed=42)
    act = torch.randint(0, 12, (steps, total, 20), dtype=torch.int32, generator=torch.Generator(device="cpu").manual_seed(7)).cuda()
    want = env.step_many(act)["ep_sums"].cpu().numpy()
    assert got.shape == (total, 5)
    np.testing.assert_array_equal(got, want)
    # the transitions rank 0 received from both ranks (gather_transitions): each row is the row of the unsharded rollout
    # its global index names
    env.reset(seed=42, episode=0)
    full = env.step_many(act)
    tr = np.load(str(tmp_path / f"gather_{backend}.npy") + ".transitions.npz")
    idx = torch.from_numpy(tr["index"]).cuda()
    assert idx.shape == (1024,) and idx.unique().numel() == 1024
    t, rem = idx // (total * 20), idx % (total * 20)
    b, i = rem // 20, rem % 20
    np.testing.assert_array_equal(tr["next_states"], full["obs"][t, b, i].cpu().numpy())
    np.testing.assert_array_equal(tr["rewards"], full["reward"][t, b, i].cpu().numpy())
    np.testing.assert_array_equal(tr["actions"], act[t, b, i].cpu().numpy())
    prev = torch.where((t == 0)[:, None], torch.full((1, 12), -1.0, device="cuda"), full["obs"][(t - 1).clamp(min=0), b, i])
    np.testing.assert_array_equal(tr["states"], prev.cpu().numpy())
    env.close()


@pytest.mark.parametrize("N,M,box", [(20, 10, 2000.0), (20, 10, 500.0), (10, 10, 700.0), (5, 3, 400.0)])
def test_pmi_long_launch_pooled_slots_equals_single_steps(uavtrack, pmi_state_dict, N, M, box):
    """MAAC-R launches of >= 16 steps run on single-wavefront groups that take their pair-list slots from a private pool
    (block reservations, UAV-granular block switches, dummy records the scorer skips); shorter launches and
    uavtrack_step use the 4-wave geometry with one reservation per workgroup-step.  Both must give the same bits: one
    40-step launch == 40 single steps (rewards, observations, terms, coverage, episode sums within rounding, final state),
    in the reference box and in the dense one (many pairs per step: blocks above the default size, frequent switches)."""
    B, T = 300, 40        # (the three swarm shapes the single-wavefront variant is built for: 3, 6 and 12 environments per wavefront)
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3, x_max=box, y_max=box,
                             reward_mode=uavtrack.RewardMode.PMI)
    a, b = uavtrack.BatchedUavEnv(cfg), uavtrack.BatchedUavEnv(cfg)
    a.set_pmi(pmi_state_dict); b.set_pmi(pmi_state_dict)
    a.reset(seed=17); b.reset(seed=17)
    assert a.kernel_info()["workgroup"] == 64            # the long-launch geometry
    act = torch.randint(0, 12, (T, B, N), dtype=torch.int32, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    p0 = a.pmi_pairs_scored()
    many = a.step_many(act)
    pairs_many = a.pmi_pairs_scored() - p0
    p0 = b.pmi_pairs_scored()
    ep = torch.zeros(B, 5, device="cuda")
    for t in range(T):
        obs, rew, _ = b.step(act[t])
        assert torch.equal(rew, many["reward"][t]), t
        assert torch.equal(obs, many["obs"][t]), t
        assert torch.equal(b.info["terms"], many["terms"][t]) and torch.equal(b.info["covered"], many["covered"][t]), t
        ep[:, 0] += rew.mean(1); ep[:, 1:4] += b.info["terms"].mean(2).T; ep[:, 4] += b.info["covered"]
    assert b.pmi_pairs_scored() - p0 == pairs_many > 0   # the same real pairs, whatever the dummies
    torch.testing.assert_close(many["ep_sums"], ep, rtol=1e-5, atol=1e-5)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert float((many["reward"] - (1 - 0.3) * 0).abs().max()) > 0      # (rewards are not all zero)
