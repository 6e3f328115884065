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
