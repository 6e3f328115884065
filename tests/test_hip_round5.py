"""GPU tests of round 5 (through the C ABI): the raw-reward output, the host-facing step and the B = 1 adapter on it,
range tests one fp32 ulp either side of their thresholds (g9, recorded from the reference), workgroups beyond 64 KiB of
LDS, MAAC-R scratch and weight uploads that fail without side effects, graph capture of MAAC-R launches."""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from oracle import OracleConfig, OracleEnv
from test_hip_parity import ATOL, MARGIN, ang_diff, host, inject

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uavtrack():
    import uavtrack
    return uavtrack


def _ref_cfg(n, m, coop=0.0):
    return {"environment": {"n_uav": n, "m_targets": m, "x_max": 2000, "y_max": 2000, "na": 12},
            "uav": {"dt": 1, "v_max": 20, "h_max": 6, "dc": 500, "dp": 200, "alpha": 0.6, "beta": 0.2, "gamma": 0.2},
            "target": {"v_max": 5, "h_max": 6}, "cooperative": coop}


def test_one_ulp_either_side_of_every_threshold(uavtrack):
    """g9 (recorded from the reference): distances ONE fp32 ulp inside / outside dp, dc and 2 dp after the move -- every
    coordinate an fp32 number before and after it, so there is nothing to set aside: each inclusive / strict decision, the
    observation rows, terms, raw and cooperative rewards and the coverage count must be the reference's."""
    z, meta = load_golden("g9_ulp_edges")
    for case in meta["cases"]:
        name, N, M = case["name"], case["n_uav"], case["m_targets"]
        g = lambda k: z[f"{name}__{k}"]
        for k in ("ux", "uy", "tx", "ty"):        # the premise: the recorded fp64 poses are fp32 numbers
            assert np.array_equal(g(k).astype(np.float32).astype(np.float64), g(k)), k
        env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=1, n_uav=N, m_targets=M, cooperative=case["cooperative"]))
        raw = torch.empty(1, 1, N, device="cuda")
        for with_raw in (False, True):            # the plain kernel variant and the one with the per-step extras
            env.set_raw_output(raw if with_raw else None)
            env.set_state(**{k: g(k)[0][None] for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th")})
            obs, rew, _ = env.step(torch.from_numpy(g("actions")[0][None].astype(np.int32)))
            np.testing.assert_allclose(obs[0].cpu().numpy(), g("obs")[0], rtol=0, atol=ATOL, err_msg=f"{name} obs")
            np.testing.assert_allclose(rew[0].cpu().numpy(), g("reward")[0], rtol=0, atol=ATOL, err_msg=f"{name} reward")
            np.testing.assert_allclose(env.info["terms"][:, 0].cpu().numpy(), g("terms")[0], rtol=0, atol=ATOL, err_msg=f"{name} terms")
            assert int(env.info["covered"][0]) == int(g("covered")[0]), name
            if with_raw:
                np.testing.assert_allclose(raw[0, 0].cpu().numpy(), g("raw")[0], rtol=0, atol=ATOL, err_msg=f"{name} raw")
            st = host(env.get_state())
            for k in ("ux", "uy", "tx", "ty"):    # the moves are exact
                np.testing.assert_array_equal(st[k][0].astype(np.float64), g(k)[1], err_msg=f"{name} {k}")
        env.close()
    # the two cases really differ where the thresholds say they must
    assert np.all(z["ulp_inside__obs"][0, 0, 5:9] != -1.0) and np.all(z["ulp_outside__obs"][0, 0, 5:9] == -1.0)


@pytest.mark.parametrize("N,M,mode,dim", [(20, 10, "raw", 2), (20, 10, "mean", 2), (20, 10, "pmi", 2), (50, 25, "pmi", 2),
                                          (7, 4, "mean", 2), (70, 5, "pmi", 2), (50, 25, "raw", 3)])
def test_raw_reward_output(uavtrack, pmi_state_dict, N, M, mode, dim):
    """uavtrack_set_raw_reward_output: uav.raw_reward (environment.py:219) of every step against the oracle's, fused ==
    stepwise bit for bit, and every other output unchanged bit for bit by asking for it (another kernel variant runs)."""
    from oracle import OraclePmi
    B, T = 48, 6
    coop = 0.0 if mode == "raw" else 0.3
    kw = dict(n_envs=B, n_uav=N, m_targets=M, cooperative=coop, dim=dim, nc=3 if dim == 3 else 1, z_max=300.0)
    rm = uavtrack.RewardMode.PMI if mode == "pmi" else None
    a, b = (uavtrack.BatchedUavEnv(uavtrack.EnvConfig(reward_mode=rm, **kw)) for _ in range(2))
    orc = OracleEnv(OracleConfig(**kw), n_threads=8)
    if mode == "pmi":
        a.set_pmi(pmi_state_dict); b.set_pmi(pmi_state_dict)
        orc.pmi = OraclePmi.from_state_dict(pmi_state_dict)
    a.reset(seed=11); b.reset(seed=11)
    na = 12 * (3 if dim == 3 else 1)
    act = torch.randint(0, na, (T, B, N), dtype=torch.int32, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    st0 = host(a.get_state())
    plain = a.step_many(act)
    a.set_state(**{k: v for k, v in st0.items()})
    fused = a.step_many(act, want_raw=True)
    assert fused["raw"].shape == (T, B, N)
    for k in ("obs", "reward", "terms", "covered", "done"):
        assert torch.equal(fused[k], plain[k]), k
    rawbuf = torch.empty(1, B, N, device="cuda")
    b.set_raw_output(rawbuf)
    inject(orc, st0)
    for t in range(T):
        inject(orc, host(b.get_state()))
        obs, rew, _ = b.step(act[t])
        assert torch.equal(obs, fused["obs"][t]) and torch.equal(rew, fused["reward"][t])
        assert torch.equal(b.info["raw"], fused["raw"][t]), t
        ref = orc.step(act[t].cpu().numpy())
        okr = ref["margin_row"] > MARGIN
        np.testing.assert_allclose(fused["raw"][t].cpu().numpy()[okr], ref["raw"][okr], rtol=0, atol=ATOL, err_msg=f"raw t{t}")
        if mode == "raw":
            assert torch.equal(fused["raw"][t], fused["reward"][t])     # MAAC: the reward IS the raw reward (|raw| <= 1)
    assert bool((fused["raw"] != fused["reward"]).any()) == (mode != "raw")
    # a buffer too small for the call is refused, and switching the output off again restores the plain variant
    with pytest.raises(RuntimeError, match="raw-reward buffer"):
        b.step_many(act)
    b.set_raw_output(None)
    a.close(); b.close()


@pytest.mark.parametrize("N,M,mode", [(5, 3, "raw"), (20, 10, "raw"), (20, 10, "mean"), (20, 10, "pmi"), (50, 25, "pmi"), (33, 7, "mean")])
def test_step_host_equals_device_step(uavtrack, pmi_state_dict, N, M, mode):
    """uavtrack_step_host (Environment.step for a host caller: host actions in, results in the library's page-locked block)
    against uavtrack_step on device buffers from the same state: every output and the state snapshot bit for bit."""
    B = 3
    coop = 0.0 if mode == "raw" else 0.3
    rm = uavtrack.RewardMode.PMI if mode == "pmi" else None
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=coop, reward_mode=rm, horizon=4)
    a, b = uavtrack.BatchedUavEnv(cfg), uavtrack.BatchedUavEnv(cfg)
    if mode == "pmi":
        a.set_pmi(pmi_state_dict); b.set_pmi(pmi_state_dict)
    a.reset(seed=5); b.reset(seed=5)
    raw = torch.empty(1, B, N, device="cuda")
    b.set_raw_output(raw)
    rng = np.random.RandomState(1)
    for t in range(6):
        act = rng.randint(0, 12, size=(B, N)).astype(np.int32)
        v = a.step_host(act)
        obs, rew, done = b.step(torch.from_numpy(act))
        np.testing.assert_array_equal(v["obs"], obs.cpu().numpy())
        np.testing.assert_array_equal(v["reward"], rew.cpu().numpy())
        np.testing.assert_array_equal(v["terms"], b.info["terms"].cpu().numpy())
        np.testing.assert_array_equal(v["raw"], raw[0].cpu().numpy())
        np.testing.assert_array_equal(v["covered"], b.info["covered"].cpu().numpy())
        np.testing.assert_array_equal(v["done"].astype(bool), done.cpu().numpy())
        assert bool(v["done"].all()) == (t + 1 >= 4)
        st = host(b.get_state())
        for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th", "step_count"):
            np.testing.assert_array_equal(v[k], st[k], err_msg=k)
        np.testing.assert_array_equal(v["ua"], act)
    # the device-side state of `a` moved with the host steps: a device step from here agrees with `b`
    act = torch.from_numpy(rng.randint(0, 12, size=(B, N)).astype(np.int32))
    oa, ra, _ = a.step(act)
    b.set_raw_output(None)
    ob, rb, _ = b.step(act)
    assert torch.equal(oa, ob) and torch.equal(ra, rb)
    with pytest.raises(ValueError):
        a.step_host(np.zeros((B, N), dtype=np.int64))
    a.close(); b.close()


def test_compat_adapter_raw_reward_and_views_vs_reference_recording(uavtrack):
    """The reference-shaped Environment on uavtrack_step_host: g1 (recorded from the reference, random.seed(42), no state
    injection) step by step -- next_states, the reward dict, covered, uav.raw_reward, uav.x / target.x and env.position."""
    z, meta = load_golden("g1_n5m3_raw")
    cfg = _ref_cfg(5, 3)
    env = uavtrack.Environment(n_uav=5, m_targets=3, x_max=2000, y_max=2000, na=12)
    random.seed(meta["seeds"][0])
    env.reset(config=cfg)
    np.testing.assert_allclose(np.array(env.get_states()), z["obs0"][0], rtol=0, atol=1e-6)
    raw_err, obs_err = [], []
    for t in range(60):
        a = [random.randint(0, 11) for _ in range(5)]
        assert a == [int(v) for v in z["actions"][0, t]]
        nxt, rd, cov = env.step(cfg, None, a)
        assert isinstance(nxt, list) and len(nxt) == 5 and nxt[0].shape == (12,) and nxt[0].dtype == np.float64
        assert isinstance(rd["rewards"], list) and isinstance(rd["rewards"][0], float) and isinstance(cov, int)
        got_raw = np.array([u.raw_reward for u in env.uav_list])
        # free-running fp32 against the recorded fp64 run: drift level, a knife-edge step may flip a range test (bounded below)
        raw_err.append(np.abs(got_raw - z["raw"][0, t]).max())
        obs_err.append(np.abs(np.array(nxt) - z["obs"][0, t]).max())
        np.testing.assert_array_equal(got_raw, np.array(rd["rewards"]))       # cooperative = 0: reward = raw (uav.py:270)
        np.testing.assert_array_equal([u.reward for u in env.uav_list], rd["rewards"])
        np.testing.assert_allclose([u.x for u in env.uav_list], z["ux"][0, t + 1], rtol=0, atol=2e-2)
        np.testing.assert_allclose([k.x for k in env.target_list], z["tx"][0, t + 1], rtol=0, atol=2e-2)
        np.testing.assert_allclose([u.h for u in env.uav_list], z["uh"][0, t + 1], rtol=0, atol=1e-4)
        assert [u.a for u in env.uav_list] == a
        assert env.position["all_uav_xs"][-1] == [u.x for u in env.uav_list]
        assert env.position["all_target_ys"][-1] == [k.y for k in env.target_list]
        for i, u in enumerate(env.uav_list):
            np.testing.assert_array_equal(u.get_local_state(), nxt[i])
    assert np.median(raw_err) < 1e-5 and np.sum(np.array(raw_err) > 1e-3) <= 2, (np.median(raw_err), np.max(raw_err))
    assert np.median(obs_err) < 1e-4
    assert len(env.position["all_uav_xs"]) == 60 and len(env.position["all_target_xs"][0]) == 3
    assert sum(int(c != int(r)) for c, r in zip(env.covered_target_num, z["covered"][0, :60])) <= 1
    # a config with MORE UAVs than the environment: start positions spaced by x_max / (n_cfg + 1) (environment.py:54-59, 105)
    random.seed(42)
    env.reset(config=_ref_cfg(6, 3))
    np.testing.assert_allclose([u.x for u in env.uav_list], [i * 2000 / 7 for i in range(1, 6)], rtol=1e-7)


def test_compat_greedy_actions_reproducible_across_processes(uavtrack):
    """the C-METHOD path of the adapter must give the same actions in every process after the same random.seed."""
    code = f"""
import sys; sys.path[:0] = {[ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]!r}
import random, uavtrack
cfg = {_ref_cfg(5, 3)!r}
env = uavtrack.Environment(n_uav=5, m_targets=3, x_max=2000, y_max=2000, na=12)
random.seed(7); env.reset(config=cfg)
out = []
for t in range(4):
    a = [u.get_action_by_direction(env.target_list, env.uav_list) for u in env.uav_list]
    out.append(a); env.step(cfg, None, a)
print(out)
"""
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip() for _ in range(2)]
    assert outs[0] == outs[1] and outs[0].startswith("[["), outs


@pytest.mark.parametrize("N,M,mode", [(512, 3000, "raw"), (64, 4096, "mean"), (300, 2000, "pmi"), (3, 4000, "raw"), (512, 4096, "pmi")])
def test_workgroups_beyond_64_kib_of_lds(uavtrack, pmi_state_dict, N, M, mode):
    """shapes whose ONE environment needs more than the 64 KiB of LDS a launch gets by default run on up
    to the CU's 160 KiB (hipFuncSetAttribute) -- teacher-forced against the oracle like every other shape."""
    from oracle import OraclePmi
    B = 5
    coop = 0.0 if mode == "raw" else 0.3
    kw = dict(n_envs=B, n_uav=N, m_targets=M, cooperative=coop)
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(reward_mode=uavtrack.RewardMode.PMI if mode == "pmi" else None, **kw))
    info = env.kernel_info()
    assert 64 * 1024 < info["lds_bytes"] <= 160 * 1024 and info["envs_per_workgroup"] == 1, info
    orc = OracleEnv(OracleConfig(**kw), n_threads=8)
    if mode == "pmi":
        env.set_pmi(pmi_state_dict)
        orc.pmi = OraclePmi.from_state_dict(pmi_state_dict)
    env.reset(seed=77)
    rng = np.random.RandomState(N)
    acts = rng.randint(0, 12, size=(3, B, N)).astype(np.int32)
    for t in range(3):
        inject(orc, host(env.get_state()))
        obs, rew, _ = env.step(torch.from_numpy(acts[t]))
        ref = orc.step(acts[t])
        ok, okr = ref["margin"] > MARGIN, ref["margin_row"] > MARGIN
        np.testing.assert_allclose(obs.cpu().numpy()[okr], ref["obs"][okr], rtol=0, atol=ATOL)
        np.testing.assert_allclose(env.info["terms"].cpu().numpy()[:, okr], ref["terms"][:, okr], rtol=0, atol=ATOL)
        rok = okr if coop == 0 else ok
        if rok.any():
            np.testing.assert_allclose(rew.cpu().numpy()[rok], ref["reward"][rok], rtol=0, atol=ATOL)
        np.testing.assert_array_equal(env.info["covered"].cpu().numpy()[ok], ref["covered"][ok])
    # fused == stepwise also on the large-LDS launch
    env.reset(seed=78)
    twin = uavtrack.BatchedUavEnv(env.cfg)
    if mode == "pmi":
        twin.set_pmi(pmi_state_dict)
    twin.reset(seed=78, episode=env._episode - 1)
    many = env.step_many(torch.from_numpy(acts))
    for t in range(3):
        o, r, _ = twin.step(torch.from_numpy(acts[t]))
        assert torch.equal(o, many["obs"][t]) and torch.equal(r, many["reward"][t])
    env.close(); twin.close()
    # one environment that cannot fit a CU's LDS at all is turned away at create, with the reason
    with pytest.raises(RuntimeError, match="LDS"):
        uavtrack.BatchedUavEnv(uavtrack.EnvConfig(n_envs=1, n_uav=512, m_targets=4096, cooperative=0.0, dim=3, nc=3))


def test_set_pmi_weights_failure_leaves_previous_weights(uavtrack, pmi_state_dict):
    """with a scorer pinned, weights it cannot take are refused BEFORE anything is replaced -- the handle keeps
    scoring with the weights it had."""
    B, N, M = 32, 20, 10
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3, reward_mode=uavtrack.RewardMode.PMI, x_max=600.0, y_max=600.0)
    a, b = uavtrack.BatchedUavEnv(cfg), uavtrack.BatchedUavEnv(cfg)
    a.set_pmi(pmi_state_dict); b.set_pmi(pmi_state_dict)
    a.set_pmi_scheme("f16x3")
    bad = {k: np.array(v, copy=True) for k, v in pmi_state_dict.items()}
    bad["fc1.weight"] = bad["fc1.weight"] * 1.0e6          # far beyond f16's range: only the bf16 / fp32 scorers take it
    with pytest.raises(RuntimeError, match="previous weights stay loaded"):
        a.set_pmi(bad)
    assert a.pmi_info()["scheme"] == "f16x3" and a.pmi_info()["f16_range_ok"]
    a.reset(seed=3); b.reset(seed=3)
    act = torch.randint(0, 12, (4, B, N), dtype=torch.int32, device="cuda", generator=torch.Generator("cuda").manual_seed(9))
    ra, rb = a.step_many(act), b.step_many(act)
    assert torch.equal(ra["reward"], rb["reward"]) and (ra["reward"] != 0).any()
    # unpinned, the same weights load (and select the wide-range kernel)
    a.set_pmi_scheme("auto")
    a.set_pmi(bad)
    assert a.pmi_info()["scheme"] == "bf16x6"
    a.close(); b.close()


def test_maac_r_calls_capture_into_a_graph_and_growth_is_refused_under_capture(uavtrack, pmi_state_dict):
    """MAAC-R scratch is sized for cfg.horizon steps when the weights are set, so a first long call
    neither synchronises nor allocates and can be captured into a HIP graph; a call LONGER than the horizon would have
    to grow it, and under capture it is refused with a message of its own (outside capture it grows)."""
    B, N, M, H = 64, 20, 10, 12
    cfg = uavtrack.EnvConfig(n_envs=B, n_uav=N, m_targets=M, cooperative=0.3, reward_mode=uavtrack.RewardMode.PMI, horizon=H,
                             x_max=700.0, y_max=700.0)
    a, b = uavtrack.BatchedUavEnv(cfg), uavtrack.BatchedUavEnv(cfg)
    a.set_pmi(pmi_state_dict); b.set_pmi(pmi_state_dict)
    a.reset(seed=1); b.reset(seed=1)
    gen = torch.Generator("cuda").manual_seed(2)
    act = torch.randint(0, 12, (H, B, N), dtype=torch.int32, device="cuda", generator=gen)
    long_act = torch.randint(0, 12, (3 * H, B, N), dtype=torch.int32, device="cuda", generator=gen)
    want = b.step_many(act)
    out = {k: torch.empty_like(v) for k, v in want.items()}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            a.step_many(act, out=out)                   # the handle's FIRST stepping call: captured, not run
            with pytest.raises(RuntimeError, match="being captured"):
                a.step_many(long_act)
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    torch.cuda.synchronize()
    for k in ("obs", "reward", "terms", "covered", "done", "ep_sums"):
        assert torch.equal(out[k], want[k]), k
    # outside capture the long call grows the scratch and agrees with stepping in two halves
    a.reset(seed=5); b.reset(seed=5)
    ra = a.step_many(long_act)
    rb = [b.step_many(long_act[k * H:(k + 1) * H]) for k in range(3)]
    assert torch.equal(ra["reward"], torch.cat([r["reward"] for r in rb]))
    a.close(); b.close()


@pytest.mark.parametrize("N,M", [(10, 10), (20, 10), (50, 25), (5, 3), (7, 4), (70, 5)])
@pytest.mark.parametrize("mode", ["raw", "mean", "pmi"])
def test_3d_ranges_use_the_altitude_in_every_reward_mode(uavtrack, pmi_state_dict, N, M, mode):
    """3-D with the swarm spread over the WHOLE altitude band (after a reset every UAV flies at z_max / 2 and a few steps of
    climbing separate them by metres only, so a range test that forgot z would pass): UAVs and targets packed into a 300 m
    square, altitudes uniform in [0, 600] -- most pairs are within dp in the plane and outside it in space.  Every range test
    of the step (observation, tracking, coverage, duplicate term, cooperative neighbours) against the oracle, all three
    reward modes, specialised and general kernels.  (Round 4 shared MAAC-R's duplicate term between the two UAVs of a pair
    and left its neighbour mask on the planar distance in 3-D: tests/fuzz_api.py found it in round 5.)"""
    from oracle import OraclePmi
    B = 40
    coop = 0.0 if mode == "raw" else 0.3
    kw = dict(n_envs=B, n_uav=N, m_targets=M, cooperative=coop, dim=3, nc=3, z_max=600.0, x_max=300.0, y_max=300.0)
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(reward_mode=uavtrack.RewardMode.PMI if mode == "pmi" else None, **kw))
    orc = OracleEnv(OracleConfig(**kw), n_threads=8)
    if mode == "pmi":
        env.set_pmi(pmi_state_dict)
        orc.pmi = OraclePmi.from_state_dict(pmi_state_dict)
    env.reset(seed=9)
    rng = np.random.RandomState(N + M)
    st = host(env.get_state())
    st["uz"] = rng.uniform(0.0, 600.0, size=(B, N)).astype(np.float32)
    st["tz"] = rng.uniform(0.0, 600.0, size=(B, M)).astype(np.float32)
    env.set_state(**st)
    planar_only = spatial = 0
    for t in range(5):
        cur = host(env.get_state())
        inject(orc, cur)
        act = rng.randint(0, 36, size=(B, N)).astype(np.int32)
        obs, rew, _ = env.step(torch.from_numpy(act))
        ref = orc.step(act)
        ok, okr = ref["margin"] > MARGIN, ref["margin_row"] > MARGIN
        rok = okr if coop == 0 else np.broadcast_to(ok[:, None], okr.shape)
        np.testing.assert_allclose(obs.cpu().numpy()[okr], ref["obs"][okr], rtol=0, atol=ATOL, err_msg=f"obs t{t}")
        np.testing.assert_allclose(env.info["terms"].cpu().numpy()[:, okr], ref["terms"][:, okr], rtol=0, atol=ATOL, err_msg=f"terms t{t}")
        np.testing.assert_allclose(rew.cpu().numpy()[rok], ref["reward"][rok], rtol=0, atol=ATOL, err_msg=f"reward t{t}")
        np.testing.assert_array_equal(env.info["covered"].cpu().numpy()[ok], ref["covered"][ok], err_msg=f"covered t{t}")
        nx = host(env.get_state())
        d2 = (nx["ux"][:, :, None] - nx["ux"][:, None]) ** 2 + (nx["uy"][:, :, None] - nx["uy"][:, None]) ** 2
        d3 = d2 + (nx["uz"][:, :, None] - nx["uz"][:, None]) ** 2
        off = ~np.eye(N, dtype=bool)[None]
        planar_only += int(((d2 <= 200.0 ** 2) & (d3 > 200.0 ** 2) & off).sum())
        spatial += int(((d3 <= 200.0 ** 2) & off).sum())
    if N > 1:
        assert planar_only > 0 and spatial > 0, (planar_only, spatial)      # the scenario separates the planar test from the spatial one
    env.close()
