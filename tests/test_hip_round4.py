"""GPU parity, round-4 additions (all through the C ABI):

* "identical seeds": the reference-shaped B = 1 adapter after `random.seed(s)` with NO state injection -- the reference's
  initial state, the reference's actions, the reference's trajectory (g1), the four reset layouts of g6;
* f4: the g1 episode through export.save_rollout / the adapter's save_position + save_covered_num against the CSV files
  the reference's own writers produced (tests/golden/f4_export.npz);
* f3 on the device: the reference's train_pmi selection on a device-resident history (tests/golden/f3_pmi_train.npz);
* every scorer the library can dispatch (f16 x 3 "t3", bf16 x 6 "x6", fp32 MFMA) under the reference's MAAC-R goldens and
  the fp64 forward; networks that trip the f16 range guard land on x6; an observation that leaves f16's range at run time
  is re-scored by the wide-range kernel;
* launch geometry per launch (MAAC-R launches that cannot use the single-wavefront variant keep 256-thread groups);
* the end-of-rollout transition gather.
"""
import io
import os
import random

import numpy as np
import pytest
import torch

from conftest import adversarial_pmi_state_dict, load_golden, pmi_forward_fp64
from oracle import OracleConfig, OracleEnv, OraclePmi
from test_hip_parity import ATOL, Tally, compare_step, host, inject

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def uavtrack():
    import uavtrack
    return uavtrack


def ref_cfg(n, m, coop=0):
    return {"environment": {"n_uav": n, "m_targets": m, "x_max": 2000, "y_max": 2000, "na": 12},
            "uav": {"dt": 1, "v_max": 20, "h_max": 6, "dc": 500, "dp": 200, "alpha": 0.6, "beta": 0.2, "gamma": 0.2},
            "target": {"v_max": 5, "h_max": 6}, "cooperative": coop}


def test_compat_seed_identical_free_run_reproduces_g1(uavtrack):
    """north_star: "outputs match ... on identical seeds".  `random.seed(42); env.reset(cfg)`, then 200 steps with actions
    the CALLER draws from `random.randint` between the steps -- the run oracle/gen_golden.py recorded from the reference as
    g1 -- and nothing is injected: the initial state is the reference's (rounded to the fp32 state), every action list is
    the reference's (the adapter has consumed what the reference's reset and steps consume from the global generator), and
    the free-running fp32 trajectory stays within the drift bound of the fp64 one."""
    z, meta = load_golden("g1_n5m3_raw")
    cfg = ref_cfg(5, 3)
    env = uavtrack.Environment(n_uav=5, m_targets=3, x_max=2000, y_max=2000, na=12)
    random.seed(meta["seeds"][0])
    assert env.reset(config=cfg) is None
    st = host(env._env.get_state())
    for k in ("ux", "uy", "uh", "tx", "ty", "th"):
        np.testing.assert_array_equal(st[k][0], z[k][0, 0].astype(np.float32), err_msg=k)
    np.testing.assert_array_equal(st["ua"][0], z["ua"][0, 0])
    np.testing.assert_allclose(np.array(env.get_states()), z["obs0"][0], rtol=0, atol=1e-12)
    T = meta["steps"]
    cov_diff, rew_err, obs_err = 0, [], []
    for t in range(T):
        a = [random.randint(0, 11) for _ in range(5)]
        assert a == [int(v) for v in z["actions"][0, t]], f"action draw of step {t} left the reference's stream"
        nxt, reward, covered = env.step(cfg, None, a)
        cov_diff += int(covered != int(z["covered"][0, t]))
        rew_err.append(np.abs(np.array(reward["rewards"]) - z["reward"][0, t]).max())
        obs_err.append(np.abs(np.array(nxt) - z["obs"][0, t]).max())
    st = host(env._env.get_state())
    for k in ("ux", "uy", "tx", "ty"):
        assert np.abs(st[k][0] - z[k][0, T]).max() < 5e-2, k            # the drift bound of test_free_running_rollout_stays_close
    np.testing.assert_array_equal(st["ua"][0], z["ua"][0, T])
    # free-running fp32 against fp64: a range test may flip on a knife-edge step (DESIGN: 0.1 % of env-steps); everything else
    # stays at drift level
    assert cov_diff <= 2, cov_diff
    assert np.median(rew_err) < 1e-5 and np.median(obs_err) < 1e-4
    assert np.sum(np.array(rew_err) > 1e-3) <= 4
    # the generator is where the reference's is: the next draw of a second episode starts from the same point
    assert len(env.covered_target_num) == T and len(env.position["all_uav_xs"]) == T


def test_compat_reset_layouts_g6_exact(uavtrack):
    """g6: the four reset layouts the reference recorded under random.seed(42), through the adapter with no injection."""
    z, meta = load_golden("g6_reset")
    for tag, mm in meta.items():
        n, m = mm["n_uav"], mm["m_targets"]
        env = uavtrack.Environment(n_uav=n, m_targets=m, x_max=2000, y_max=2000, na=12)
        random.seed(42)
        env.reset(config=ref_cfg(n, m))
        st = host(env._env.get_state())
        for k in ("ux", "uy", "uh", "tx", "ty", "th"):
            np.testing.assert_array_equal(st[k][0], z[f"{tag}_{k}"].astype(np.float32), err_msg=f"{tag} {k}")
        np.testing.assert_array_equal(st["ua"][0], z[f"{tag}_ua"])
        np.testing.assert_allclose(np.array([u.get_local_state() for u in env.uav_list]), z[f"{tag}_obs0"], rtol=0, atol=1e-12)
        assert abs(env.uav_list[1].x - float(z[f"{tag}_ux"][1])) < 1e-3 and abs(env.target_list[0].y - float(z[f"{tag}_ty"][0])) < 1e-3
        env._env.close()


def _parse_csv(raw: bytes):
    lines = raw.decode().strip().split("\n")
    return lines[0], np.array([[float(v) for v in ln.split(",")] for ln in lines[1:]])


def test_export_csv_against_reference_writers(uavtrack, tmp_path):
    """f4 against the reference itself: tests/golden/f4_export.npz holds the three files
    Environment.save_position / save_covered_num (environment.py:229-244) wrote for the g1 episode.  The same episode,
    teacher-forced on the device (g1's recorded state before every step, fp32-rounded; g1's actions), exported (a) from
    the batched outputs by export.save_rollout and (b) by the B = 1 adapter's own writers: covered_target_num byte for
    byte, the pose files line for line with values equal to fp32 rounding of a 2000 m coordinate."""
    from uavtrack.export import save_rollout
    z, meta = load_golden("g1_n5m3_raw")
    f4, fmeta = load_golden("f4_export")
    ep = fmeta["epoch_i"]
    T, N, M = meta["steps"], 5, 3
    # (a) the T steps as ONE batch of T single-step environments: env t holds the state before step t
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=T, n_uav=N, m_targets=M))
    env.set_state(**{k: z[k][0, :T] for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")})
    res = env.step_many(torch.from_numpy(z["actions"][0].astype(np.int32))[None].cuda(), want_targets=True)
    # -> the layout of a T-step rollout of one environment: [T, 1, ...]
    roll = dict(obs=res["obs"][0][:, None], targets=res["targets"][0][:, None], covered=res["covered"][0][:, None])
    d1 = tmp_path / "batched"
    paths = save_rollout(str(d1), ep, roll, dc=500.0, env_index=0)
    assert [os.path.basename(p) for p in paths] == [f"u_xy{ep}.csv", f"t_xy{ep}.csv", f"covered_target_num{ep}.csv"]
    # (b) the adapter, stepped with the same injected states
    one = uavtrack.Environment(n_uav=N, m_targets=M, x_max=2000, y_max=2000, na=12)
    cfg = ref_cfg(N, M)
    one.reset(cfg)
    for t in range(T):
        one._env.set_state(**{k: z[k][0, t][None] for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")})
        one.step(cfg, None, [int(a) for a in z["actions"][0, t]])
    d2 = tmp_path / "adapter"
    for sub in ("u_xy", "t_xy", "covered_target_num"):
        (d2 / sub).mkdir(parents=True)
    one.save_position(str(d2), ep)
    one.save_covered_num(str(d2), ep)
    for d in (d1, d2):
        got = open(d / "covered_target_num" / f"covered_target_num{ep}.csv", "rb").read()
        assert got == f4["covered_target_num"].tobytes(), f"{d.name}: covered_target_num differs from the reference's file"
        for sub in ("u_xy", "t_xy"):
            hdr, vals = _parse_csv(open(d / sub / f"{sub}{ep}.csv", "rb").read())
            rhdr, rvals = _parse_csv(f4[sub].tobytes())
            assert hdr == rhdr == "x,y" and vals.shape == rvals.shape == ((N if sub == "u_xy" else M) * T, 2)
            # one fp32 step from the fp32-rounded recorded state; u_xy of the batched form also went through x / dc * dc
            np.testing.assert_allclose(vals, rvals, rtol=0, atol=6e-4, err_msg=f"{d.name} {sub}")


def test_pmi_training_selection_on_device_against_reference(uavtrack):
    """f3 on the device: the history of tests/golden/f3_pmi_train.npz resident in HBM; under the reference's seed the
    gather returns the reference's batches (PMINet.py:78-92) and a device-resident network trained by train_pmi_epoch from
    the reference's initial weights follows the reference's recorded outputs."""
    from uavtrack import make_pmi_net, pmi_batches, sample_pmi_pairs, train_pmi_epoch
    z, meta = load_golden("f3_pmi_train")
    N, b2, bs = meta["n_uav"], meta["b2_size"], meta["batch_size"]
    data = torch.from_numpy(z["train_data"]).cuda()
    torch.manual_seed(meta["torch_seed"])
    sel, _, _ = sample_pmi_pairs(data.view(meta["steps"], N, 12), N, b2)
    assert sel.is_cuda
    for k, (x12, x13) in enumerate(pmi_batches(sel, bs)):
        np.testing.assert_array_equal(x12.cpu().numpy(), z["in_1_2"][k])
        np.testing.assert_array_equal(x13.cpu().numpy(), z["in_1_3"][k])
    torch.manual_seed(11)
    net = make_pmi_net(meta["hidden"])              # (initialised on the host, like the reference, then moved)
    net = net.cuda()
    opt = torch.optim.Adam(net.parameters(), lr=0.001)
    seen = []
    hook = net.register_forward_hook(lambda m, i, o: seen.append(o.detach().float().cpu().numpy()))
    torch.manual_seed(meta["torch_seed"])
    avg = train_pmi_epoch(net, opt, data, N, b2, bs)
    hook.remove()
    assert abs(avg - float(z["avg_loss"])) < 1e-4
    for k in range(b2 // bs):
        np.testing.assert_allclose(seen[2 * k], z["out_1_2"][k], rtol=0, atol=2e-4)
        np.testing.assert_allclose(seen[2 * k + 1], z["out_1_3"][k], rtol=0, atol=2e-4)


# ---------------------------------------------------------------------------------------------------------------------
# every scorer the library can dispatch.  Reference: PMINetwork.forward / inference, PMINet.py:41-72.

def _scaled_identity(sd, how):
    """The SAME function as `sd` (ReLU is positively homogeneous; the factors are powers of two, so the BatchNorm fold
    scales exactly), but with operands that trip the host-side f16 range guard of uavtrack_set_pmi_weights:
    "big_fc1": folded fc1 x 2^20 (|w| far past 32 000), fc2 x 2^-20;  "big_branch": folded branch layers x 2^14 (activation
    bound far past 32 000), fc1's input columns x 2^-14."""
    out = {k: np.array(v, dtype=np.float32) for k, v in sd.items()}
    if how == "big_fc1":
        s = np.float32(2.0 ** 20)
        out["bn1.weight"] *= s; out["bn1.bias"] *= s
        out["fc2.weight"] /= s
    elif how == "big_branch":
        s = np.float32(2.0 ** 14)
        for bn in ("bn_comm", "bn_obs", "bn_boundary_state"):
            out[bn + ".weight"] *= s; out[bn + ".bias"] *= s
        out["fc1.weight"] /= s
    else:
        raise ValueError(how)
    return out


SCHEME_CASES = [("auto", None, "f16x3"), ("bf16x6", None, "bf16x6"), ("fp32", None, "fp32"),
                ("auto", "big_fc1", "bf16x6"), ("auto", "big_branch", "bf16x6")]


@pytest.mark.parametrize("pin,trip,expect", SCHEME_CASES)
def test_pmi_goldens_on_every_scorer(uavtrack, pmi_state_dict, pmi_state_dict_h64, pin, trip, expect):
    """g4 (H = 128) and g4b (H = 64), the reference's recorded MAAC-R rewards, on each scorer kernel: the default f16 x 3,
    bf16 x 6 and fp32 MFMA pinned through uavtrack_set_pmi_scheme, and bf16 x 6 reached the way production reaches it --
    the f16 range guard turning the network away (the same network function with one layer scaled by a power of two and the
    next one scaled back).  uavtrack_pmi_info says which kernel the handle launches."""
    for name in ("g4_n20m10_pmi", "g4b_n20m10_pmi_h64"):
        z, meta = load_golden(name)
        sd = pmi_state_dict_h64 if name.endswith("h64") else pmi_state_dict
        if trip:
            sd = _scaled_identity(sd, trip)
        N, M = meta["n_uav"], meta["m_targets"]
        E, T = len(meta["seeds"]), meta["steps"]
        B = E * T
        pick = lambda k: z[k][:, :T].reshape(B, -1)
        kw = dict(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3)
        env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(reward_mode=uavtrack.RewardMode.PMI, **kw))
        env.set_pmi_scheme(pin)
        env.set_pmi(sd)
        info = env.pmi_info()
        assert info["scheme"] == expect, info
        assert info["f16_range_ok"] == (trip is None)
        env.set_state(**{k: pick(k) for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")})
        act = z["actions"].reshape(B, N).astype(np.int32)
        orc = OracleEnv(OracleConfig(**kw), n_threads=8)
        orc.pmi = OraclePmi.from_state_dict(sd)
        inject(orc, host(env.get_state()))
        ref = orc.step(act)
        ok = ref["margin"] > 5e-3
        obs, rew, _ = env.step(torch.from_numpy(act))
        rew = rew.cpu().numpy()
        np.testing.assert_allclose(rew[ok], z["reward"].reshape(B, N)[ok], rtol=0, atol=2e-5, err_msg=f"{name} {pin} {trip}")
        np.testing.assert_allclose(rew[ok], ref["reward"][ok], rtol=0, atol=ATOL, err_msg=f"{name} {pin} {trip} vs oracle")
        assert env.pmi_info()["rescored_chunks"] == 0          # nothing left f16's range: the stand-by kernel stayed idle
        env.close()


@pytest.mark.parametrize("pin,trip,expect", SCHEME_CASES)
def test_pmi_inference_on_every_scorer(uavtrack, pmi_state_dict, pin, trip, expect):
    """PMINetwork.inference alone (uavtrack_pmi_inference) on each scorer kernel against the fp64 forward: the adversarial
    weights (every 3H-term sum a large cancellation) at 1e-5 of sum |terms|, the reference-initialised network at the plain
    1e-5, and the tile loop's edges (fewer pairs than a tile, one more than a tile, fewer tiles than workgroups, none)."""
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=1, n_uav=2, m_targets=1, cooperative=0.3,
                                                    reward_mode=uavtrack.RewardMode.PMI))
    env.set_pmi_scheme(pin)
    rng = np.random.RandomState(11)
    for hidden in (128, 64):
        sd = adversarial_pmi_state_dict(hidden, seed=hidden)
        if trip:
            sd = _scaled_identity(sd, trip)
        env.set_pmi(sd)
        assert env.pmi_info()["scheme"] == expect
        n = 3000 + hidden
        x = rng.uniform(-1.0, 1.0, (n, 12)).astype(np.float32)
        x[: n // 4] *= rng.uniform(0.0, 4.5, (n // 4, 1)).astype(np.float32) ** 2
        got = env.pmi_inference(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
        ref, mag = pmi_forward_fp64(sd, x, want_scale=True)
        # (the scaled-identity networks carry the factor 2^20 / 2^14 in their intermediate sums; the bound scales with them)
        assert np.isfinite(got).all() and (np.abs(got - ref) <= 1e-5 * np.maximum(mag, 1.0)).all(), (hidden, np.abs(got - ref).max())
    sd = _scaled_identity(pmi_state_dict, trip) if trip else pmi_state_dict
    env.set_pmi(sd)
    for n in (1, 31, 32, 33, 65, 257, 8191):
        x = rng.uniform(-1.0, 1.0, (n, 12)).astype(np.float32)
        got = env.pmi_inference(torch.from_numpy(x).cuda()).cpu().numpy().astype(np.float64)
        ref = pmi_forward_fp64(sd, x)
        assert got.shape == (n,) and np.abs(got - ref).max() < 1e-5, (n, np.abs(got - ref).max())
    assert env.pmi_inference(torch.empty(0, 12, device="cuda")).shape == (0,)
    env.close()


def test_pmi_scheme_pinning_errors(uavtrack, pmi_state_dict):
    """A pinned scheme the weights cannot run on is an error where it is pinned or where the weights arrive, never a silent
    switch: f16 x 3 with a network beyond f16's range, the split kernels with a width they are not built for."""
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=1, n_uav=2, m_targets=1, cooperative=0.3,
                                                    reward_mode=uavtrack.RewardMode.PMI))
    assert env.pmi_info()["scheme"] is None
    env.set_pmi(_scaled_identity(pmi_state_dict, "big_fc1"))
    with pytest.raises(RuntimeError, match="f16"):
        env.set_pmi_scheme("f16x3")
    assert env.pmi_info()["scheme"] == "bf16x6"
    env.set_pmi(pmi_state_dict)
    env.set_pmi_scheme("f16x3")
    with pytest.raises(RuntimeError, match="pinned"):
        env.set_pmi(_scaled_identity(pmi_state_dict, "big_branch"))
    env.set_pmi_scheme("auto")
    from test_hip_parity import random_pmi_state_dict
    env.set_pmi(random_pmi_state_dict(200, 1))
    assert env.pmi_info() == dict(scheme="fp32", hidden_padded=224, f16_range_ok=False, rescored_chunks=0)
    with pytest.raises(RuntimeError, match="widths"):
        env.set_pmi_scheme("bf16x6")
    with pytest.raises(ValueError):
        env.set_pmi_scheme("h3")
    env.close()


def test_pmi_f16_range_watch_rescoring_near_origin(uavtrack, pmi_state_dict):
    """the uav.py:165 weight 1 / min(d, 1) is unbounded next to the origin, so observations -- and the scorer's
    inputs la_i * la_j -- can leave the range the host-side f16 guard assumed.  (1) The scorer alone on inputs up to 2e6:
    the f16 kernel must notice (uavtrack_pmi_info counts the chunk) and the scores must be the wide-range kernel's, i.e.
    within the fp32 bound of the fp64 forward; ordinary inputs leave the counter alone.  (2) A MAAC-R step with two UAVs
    a few micrometres apart, ten micrometres from the origin: weighted-mean observation rows ~10^5 times the nominal
    size, products ~10^9 (at a millimetre the products reach ~10^3, which the f16 planes' headroom still takes) -- the rewards equal those of a handle pinned to bf16 x 6 bit for bit, and every other environment
    of the batch still agrees with the oracle."""
    B, N, M = 8, 20, 10
    kw = dict(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3)
    cfg = uavtrack.EnvConfig(reward_mode=uavtrack.RewardMode.PMI, **kw)
    a, b = uavtrack.BatchedUavEnv(cfg), uavtrack.BatchedUavEnv(cfg)
    b.set_pmi_scheme("bf16x6")
    for e in (a, b):
        e.set_pmi(pmi_state_dict)
        e.reset(seed=3)
    assert a.pmi_info()["scheme"] == "f16x3"
    # (1) the network alone
    rng = np.random.RandomState(5)
    x = rng.uniform(-1.0, 1.0, (4000, 12)).astype(np.float32)
    got = a.pmi_inference(torch.from_numpy(x).cuda()).cpu().numpy()
    assert a.pmi_info()["rescored_chunks"] == 0 and np.abs(got - pmi_forward_fp64(pmi_state_dict, x)).max() < 1e-5
    xl = x.copy()
    xl[100:140] *= rng.uniform(1e3, 2e6, (40, 1)).astype(np.float32)          # a few rows far outside f16's reach
    got = a.pmi_inference(torch.from_numpy(xl).cuda()).cpu().numpy().astype(np.float64)
    ref, mag = pmi_forward_fp64(pmi_state_dict, xl, want_scale=True)
    assert a.pmi_info()["rescored_chunks"] == 1
    assert np.isfinite(got).all() and (np.abs(got - ref) <= 1e-5 * np.maximum(mag, 1.0)).all(), np.abs(got - ref).max()
    np.testing.assert_array_equal(got, b.pmi_inference(torch.from_numpy(xl).cuda()).cpu().numpy().astype(np.float64))
    # (2) the step.  Environment 2: UAVs 0 and 1 END their move next to the origin and to each other
    st = host(a.get_state())
    act = rng.randint(0, 12, size=(B, N)).astype(np.int32)
    # (a third UAV ends 60 m away: two UAVs that are each other's ONLY neighbour need no score -- their softmax weight is 1 --
    #  and the rollout kernel does not emit such a pair at all, step_kernel.hip drop_isolated)
    for i, (fx, fy, h) in enumerate(((1.0e-5, 1.0e-5, 0.7), (1.4e-5, 0.6e-5, 0.8), (50.0, 33.0, -0.4))):
        st["uh"][2, i] = h
        st["ux"][2, i] = np.float32(fx - 20.0 * np.cos(np.float32(h)))
        st["uy"][2, i] = np.float32(fy - 20.0 * np.sin(np.float32(h)))
    for e in (a, b):
        e.set_state(**{k: v for k, v in st.items() if k not in ("step_count", "episode")})
    orc = OracleEnv(OracleConfig(**kw), n_threads=4)
    orc.pmi = OraclePmi.from_state_dict(pmi_state_dict)
    inject(orc, host(a.get_state()))
    ref = orc.step(act)
    obs_a, rew_a, _ = a.step(torch.from_numpy(act))
    obs_b, rew_b, _ = b.step(torch.from_numpy(act))
    pos = host(a.get_state())
    assert max(abs(pos["ux"][2, 0]), abs(pos["uy"][2, 0]), abs(pos["ux"][2, 1]), abs(pos["uy"][2, 1])) < 3e-5
    big = np.abs(obs_a.cpu().numpy()[2, :2, :9]).max()
    assert big > 3.0e3, f"the scenario did not produce an out-of-range observation (max |obs| {big})"
    assert a.pmi_info()["rescored_chunks"] == 2 and b.pmi_info()["rescored_chunks"] == 0
    assert torch.equal(rew_a, rew_b) and torch.equal(obs_a, obs_b)
    assert torch.isfinite(rew_a).all()
    ok = ref["margin"] > 1e-3
    ok[2] = False               # (its scores hang on weights of ~10^5 +- 10 %: fp32 poses against fp64 ones, not comparable)
    assert ok.sum() >= 5
    np.testing.assert_allclose(rew_a.cpu().numpy()[ok], ref["reward"][ok], rtol=0, atol=ATOL)
    # the flag is per chunk: the next, ordinary step is scored by the f16 kernel alone again
    a.reset(seed=4)
    a.step(torch.from_numpy(act))
    assert a.pmi_info()["rescored_chunks"] == 2
    a.close(); b.close()


def test_maac_r_launch_geometry_follows_the_kernel_variant(uavtrack, pmi_state_dict):
    """a MAAC-R launch of 16+ steps uses the single-wavefront geometry only where the kernel variant
    built for it (pair-list slots from a pool) exists -- every output requested, no extras.  A launch without the terms, or
    with the automatic reset or the target trace, keeps 256-thread groups (the 4-wave emission path on 64-thread groups
    makes one global reservation per workgroup-step on four times the groups) -- and all of them give the same bits."""
    B, N, M, T = 512, 20, 10, 20
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3, reward_mode=uavtrack.RewardMode.PMI, horizon=50)
    act = torch.randint(0, 12, (T, B, N), dtype=torch.int32, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    outs = {}
    for label, kwargs in (("allout", {}), ("no_terms", dict(want_terms=False)), ("trace", dict(want_targets=True)),
                          ("autoreset", dict(auto_reset_seed=5)), ("short", None)):
        env = uavtrack.BatchedUavEnv(cfg)
        env.set_pmi(pmi_state_dict)
        env.reset(seed=8)
        assert env.kernel_info()["workgroup"] == 64           # the handle's own (small-grid) geometry
        if kwargs is None:
            res = env.step_many(act[:8])
        else:
            res = env.step_many(act, **kwargs)
        li = env.launch_info()
        if label == "allout":
            assert li["workgroup"] == 64 and li["single_wavefront_variant"] == 1, li
        else:
            assert li["workgroup"] == 256 and li["single_wavefront_variant"] == 0, (label, li)
        outs[label] = res
        env.close()
    for label in ("no_terms", "trace", "autoreset"):
        assert torch.equal(outs[label]["reward"], outs["allout"]["reward"]), label
        assert torch.equal(outs[label]["obs"], outs["allout"]["obs"]), label
    assert torch.equal(outs["short"]["reward"], outs["allout"]["reward"][:8])


def test_lds_need_follows_reward_mode(uavtrack):
    """the symmetric duplicate term's LDS region exists in the MAAC reward mode only; MAAC-G / MAAC-R handles
    do not reserve it (more environments per workgroup, larger swarms fit)."""
    raw = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=64, n_uav=50, m_targets=25))
    mean = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=64, n_uav=50, m_targets=25, cooperative=0.3))
    ki_r, ki_m = raw.kernel_info(), mean.kernel_info()
    assert ki_r["envs_per_workgroup"] == ki_m["envs_per_workgroup"] and ki_m["lds_bytes"] < ki_r["lds_bytes"]
    raw.close(); mean.close()


@pytest.mark.parametrize("n,m,shape", [(20, 10, "specialised"), (7, 4, "generic"), (50, 25, "specialised"), (40, 6, "generic")])
def test_maac_r_isolated_pairs_are_not_scored(uavtrack, pmi_state_dict, n, m, shape):
    """uav.py:287-288: the softmax over ONE neighbour is 1 whatever its score, so two UAVs that are each other's only
    neighbour need no score; the rollout kernel does not emit such a pair and the mix kernel does not read one.  Hand-placed
    post-move poses -- an isolated pair, a chain of three (both of its pairs are scored: the middle UAV has two neighbours)
    and a triangle -- in one 16-step launch (single-wavefront variant with pooled slots where the shape has one) and in
    single steps (4-wave emission), against the oracle; the count of scored pairs is exactly chain + triangle = 5 per step
    while the groups stay together."""
    B, T = 6, 16
    kw = dict(n_envs=B, n_uav=n, m_targets=m, cooperative=0.3)
    cfg = uavtrack.EnvConfig(reward_mode=uavtrack.RewardMode.PMI, **kw)
    a, b = uavtrack.BatchedUavEnv(cfg), uavtrack.BatchedUavEnv(cfg)
    orc = OracleEnv(OracleConfig(**kw), n_threads=4)
    orc.pmi = OraclePmi.from_state_dict(pmi_state_dict)
    for e in (a, b):
        e.set_pmi(pmi_state_dict)
        e.reset(seed=11)
    st = host(a.get_state())
    # every UAV far from every other one (600 m grid), then three groups: {0,1} 80 m apart; {2,3,4} in a row 150 m apart
    # (2-4 are 300 m apart: not neighbours); {5,6,...} -- a triangle with 100 m sides when the swarm has 8 or more UAVs
    for e in range(B):
        for i in range(n):
            st["ux"][e, i] = 100.0 + 450.0 * (i % 4); st["uy"][e, i] = 100.0 + 450.0 * (i // 4); st["uh"][e, i] = 0.3
        st["ux"][e, 1] = st["ux"][e, 0] + 80.0; st["uy"][e, 1] = st["uy"][e, 0]
        st["ux"][e, 2], st["uy"][e, 2] = 1000.0, 1680.0
        st["ux"][e, 3], st["uy"][e, 3] = 1150.0, 1680.0
        st["ux"][e, 4], st["uy"][e, 4] = 1300.0, 1680.0
        if n >= 8:
            st["ux"][e, 5], st["uy"][e, 5] = 250.0, 1700.0
            st["ux"][e, 6], st["uy"][e, 6] = 350.0, 1700.0
            st["ux"][e, 7], st["uy"][e, 7] = 300.0, 1786.0
    for e in (a, b):
        e.set_state(**{k: v for k, v in st.items() if k not in ("step_count", "episode")})
    inject(orc, host(a.get_state()))
    # every UAV takes the same action (5 on even steps, 6 on odd ones: -/+ the smallest turn rate): the swarm translates rigidly
    straight = np.broadcast_to(np.where(np.arange(T) % 2 == 0, 5, 6)[:, None, None], (T, B, n)).astype(np.int32).copy()
    p0 = a.pmi_pairs_scored()
    fused = a.step_many(torch.from_numpy(straight).cuda())
    scored_a = a.pmi_pairs_scored() - p0
    p0 = b.pmi_pairs_scored()
    worst, groups_ok, isolated = 0.0, 0, 0
    for t in range(T):
        ref = orc.step(straight[t])
        obs, rew, _ = b.step(torch.from_numpy(straight[t]))
        assert torch.equal(rew, fused["reward"][t]) and torch.equal(obs, fused["obs"][t])
        ok = ref["margin"] > 1e-3
        worst = max(worst, np.abs(rew.cpu().numpy() - ref["reward"])[ok].max())
        inject(orc, host(b.get_state()))
    assert worst < ATOL, worst
    scored_b = b.pmi_pairs_scored() - p0
    # what the oracle's neighbour sets say: pairs minus the isolated ones, summed over the launch
    inject(orc, {k: v for k, v in st.items()})
    want = 0
    for t in range(T):
        ref = orc.step(straight[t])
        pos = orc.get_state()
        x, y = pos["ux"], pos["uy"]
        d2 = (x[:, :, None] - x[:, None, :]) ** 2 + (y[:, :, None] - y[:, None, :]) ** 2
        nb = (d2 <= 200.0 ** 2) & ~np.eye(n, dtype=bool)[None]
        deg = nb.sum(-1)
        pairs = np.triu(nb, 1)
        iso = pairs & (deg[:, :, None] == 1) & (deg[:, None, :] == 1)
        want += int((pairs & ~iso).sum())
        isolated += int(iso.sum())
        groups_ok += int(pairs.sum() == B * (6 if n >= 8 else 3))
    assert groups_ok == T and isolated == B * T       # the hand-placed groups held together: one isolated pair per environment-step
    assert scored_a == scored_b == want == B * T * (5 if n >= 8 else 2)
    assert a.kernel_info()["specialised"] == (1 if shape == "specialised" else 0)


def test_knife_edge_census_inside_the_suite(uavtrack):
    """Every teacher-forced comparison sets aside the results whose range tests sit within 0.25 mm of a threshold (fp64
    margin): there the fp32 state the device integrates and the fp64 state the oracle integrates from the same fp32 poses may
    legitimately fall on different sides.  A UAV's observation row and (MAAC) reward are set aside on that UAV's OWN tests
    (oracle margin_row), the coverage count and the cooperative rewards on any test of the environment (margin).  This is
    the bounded form of tests/soak.py's census: how many of the results that are set aside REALLY differ from the oracle
    (covered count, any observation beyond 1e-5, any reward beyond 1e-5) -- a handful per hundred thousand, never a
    systematic effect; and nothing at all differs outside the margin."""
    from test_hip_parity import MARGIN
    shapes = [dict(B=4096, N=20, M=10, coop=0.3, box=2000.0, steps=40, dim=2),
              dict(B=1024, N=20, M=10, coop=0.0, box=300.0, steps=40, dim=2),       # tiny box: everything in range, reflections
              dict(B=128, N=50, M=25, coop=0.0, box=2000.0, steps=20, dim=3)]
    aside = total = really = 0                      # UAV rows
    env_aside = env_total = env_really = 0          # environments
    for k, s in enumerate(shapes):
        kw = dict(n_envs=s["B"], n_uav=s["N"], m_targets=s["M"], cooperative=s["coop"], x_max=s["box"], y_max=s["box"],
                  dim=s["dim"], nc=3 if s["dim"] == 3 else 1, z_max=300.0)
        env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(**kw))
        env.reset(seed=20 + k)
        orc = OracleEnv(OracleConfig(**kw), n_threads=8)
        rng = np.random.RandomState(k)
        na = 12 * (3 if s["dim"] == 3 else 1)
        for t in range(s["steps"]):
            inject(orc, host(env.get_state()))
            act = rng.randint(0, na, size=(s["B"], s["N"])).astype(np.int32)
            obs, rew, _ = env.step(torch.from_numpy(act))
            ref = orc.step(act)
            ok, okr = ref["margin"] > MARGIN, ref["margin_row"] > MARGIN
            o, r, cv = obs.cpu().numpy(), rew.cpu().numpy(), env.info["covered"].cpu().numpy()
            d_cov = cv != ref["covered"]                                                               # [B]
            d_obs = (np.abs(o - ref["obs"]) / (1.0 + np.abs(ref["obs"]))).max(-1) > 1e-5               # [B, N]
            d_rew = np.abs(r - ref["reward"]) > 1e-5                                                   # [B, N]
            if s["coop"] == 0:
                row_diff, env_diff = d_obs | d_rew, d_cov
            else:
                row_diff, env_diff = d_obs, d_cov | d_rew.any(1)
            assert not row_diff[okr].any(), (k, t, int(row_diff[okr].sum()))      # outside the margin: nothing differs, ever
            assert not env_diff[ok].any(), (k, t, int(env_diff[ok].sum()))
            aside += int((~okr).sum()); total += okr.size; really += int(row_diff[~okr].sum())
            env_aside += int((~ok).sum()); env_total += s["B"]; env_really += int(env_diff[~ok].sum())
        env.close()
    if os.environ.get("UAVTRACK_TEST_REPORT"):
        print(f"census: {total} UAV-steps, {aside} set aside, {really} of those really differ; "
              f"{env_total} env-steps, {env_aside} set aside, {env_really} really differ")
    assert aside <= 0.003 * total and env_aside <= 0.03 * env_total          # the exclusion itself stays small ...
    assert really + env_really <= max(3, 2e-4 * env_total), (really, env_really, aside, env_aside)      # ... and what hides inside it is a handful
