#!/usr/bin/env python3
"""Random-shape fuzzing of the fused / policy / MAAC-R entry points (GPU box; run by hand:
`python tests/fuzz_api.py [cases [seed]]`, not collected by pytest).  Per case, with random N, M, B, box, mode:
  1. uavtrack_step_many(T)        == T x uavtrack_step, bitwise (all modes incl. MAAC-R with several chunks)
  2. uavtrack_run_greedy(T)       == T x (greedy_actions, step), bitwise; greedy actions == oracle outside margins
  3. uavtrack_run_actor(T)        == T x (actor_actions, step), bitwise; probabilities == oracle within 1e-5
  4. MAAC-R teacher-forced step   == fp64 oracle within 1e-5 (random PMI weights, H in {64, 128})
  5. a shard of the batch         == the same environments of the unsharded batch, bitwise
  6. uavtrack_step_host / raw output == uavtrack_step from the same state, bitwise; raw rewards == oracle within 1e-5
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402
import uavtrack  # noqa: E402
from oracle import OracleConfig, OracleEnv, OraclePmi, actor_actions, greedy_actions  # noqa: E402
from test_hip_parity import random_pmi_state_dict  # noqa: E402


def host(d):
    return {k: v.cpu().numpy() for k, v in d.items()}


def equal_dicts(a, b, keys, what):
    for k in keys:
        assert torch.equal(a[k], b[k]), f"{what}: {k} differs"


def case(c, rng):
    N = int(rng.choice([1, 2, 3, 5, 7, 10, 16, 19, 20, 21, 33, 50, 64, 65, 90]))
    M = int(rng.choice([1, 2, 3, 5, 10, 17, 25, 33, 40, 65]))
    B = int(rng.choice([1, 3, 17, 70, 129]))
    if N * B > 6000:
        B = max(1, 6000 // N)
    box = float(rng.choice([150.0, 600.0, 2000.0]))
    mode = int(rng.choice([0, 1, 2]))                       # RAW, MEAN, PMI
    H = int(rng.choice([64, 128, 128, 32, 96, 100, 160]))     # (other widths are padded to the scorer's 32-column blocks)
    T = int(rng.choice([1, 2, 7, 13, 20, 33]))     # (>= 16: MAAC-R launches on single-wavefront groups take pooled pair-list slots)
    off = int(rng.choice([0, 5, 10 ** 10]))
    dim = int(rng.choice([2, 2, 2, 3]))
    wgs = int(rng.choice([0, 0, 64, 128, 256, 512]))          # 0: the library's own choice
    if wgs:
        os.environ["UAVTRACK_WGS"] = str(wgs)
    else:
        os.environ.pop("UAVTRACK_WGS", None)
    na_turn = int(rng.choice([12, 12, 12, 4, 7, 9]))        # environment.na (reference: 12)
    na = na_turn * (3 if dim == 3 else 1)
    tag = f"case {c}: N{N} M{M} B{B} box{box} mode{mode} H{H} T{T} off{off} dim{dim} wgs{wgs} na{na_turn}"
    kw = dict(n_envs=B, n_uav=N, m_targets=M, x_max=box, y_max=box, cooperative=0.0 if mode == 0 else 0.3,
              reward_mode=uavtrack.RewardMode(mode), env_offset=off, dim=dim, nc=3 if dim == 3 else 1, z_max=300.0, na=na_turn)
    pmi_sd = random_pmi_state_dict(H, c)
    if mode == 2:
        os.environ["UAVTRACK_PMI_SCRATCH_MB"] = str(int(rng.choice([1, 8, 2048])))

    def make(**over):
        e = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(**{**kw, **over}))
        if mode == 2:
            e.set_pmi({k: torch.from_numpy(v) for k, v in pmi_sd.items()})
        return e
    def spread(*envs):      # 3-D: a reset leaves every UAV at z_max / 2; half of the cases spread the swarm over the altitude band
        if dim == 3 and c % 2 == 0:
            r2 = np.random.RandomState(c)
            uz = torch.from_numpy(r2.uniform(0.0, 300.0, size=(B, N)).astype(np.float32)).cuda()
            tz = torch.from_numpy(r2.uniform(0.0, 300.0, size=(B, M)).astype(np.float32)).cuda()
            for e in envs:
                st = e.get_state()
                st["uz"], st["tz"] = uz.clone(), tz.clone()
                e.set_state(**st)
    a, b = make(), make()
    a.reset(seed=c); b.reset(seed=c)
    spread(a, b)
    acts = torch.from_numpy(rng.randint(0, na, size=(T, B, N)).astype(np.int32)).cuda()
    # 1. fused == single steps
    fused = a.step_many(acts)
    ep = torch.zeros(B, 5, device="cuda")
    for t in range(T):
        obs, rew, done = b.step(acts[t], ep_sums=ep)
        assert torch.equal(obs, fused["obs"][t]) and torch.equal(rew, fused["reward"][t]), f"{tag}: step_many t{t}"
        assert torch.equal(b.info["covered"], fused["covered"][t]) and torch.equal(b.info["terms"], fused["terms"][t]), tag
    equal_dicts(a.get_state(), b.get_state(), a.get_state().keys(), tag + " state")
    np.testing.assert_allclose(fused["ep_sums"].cpu().numpy(), ep.cpu().numpy(), rtol=1e-5, atol=1e-5, err_msg=tag)
    # 4. teacher-forced vs oracle (one more step from the current state)
    orc = OracleEnv(OracleConfig(n_envs=B, n_uav=N, m_targets=M, x_max=box, y_max=box, cooperative=kw["cooperative"],
                                 dim=dim, nc=3 if dim == 3 else 1, z_max=300.0, na=na_turn), n_threads=8)
    if mode == 2:
        orc.pmi = OraclePmi.from_state_dict(pmi_sd)
    st = host(a.get_state())
    orc.set_state(st["ux"], st["uy"], st["uh"], st["ua"], st["tx"], st["ty"], st["th"], uz=st.get("uz"), tz=st.get("tz"))
    act = rng.randint(0, na, size=(B, N)).astype(np.int32)
    obs, rew, _ = a.step(torch.from_numpy(act))
    ref = orc.step(act)
    ok = ref["margin"] > 2.5e-4
    if ok.any():
        err_o = np.abs(obs.cpu().numpy() - ref["obs"])[ok] / (1.0 + np.abs(ref["obs"][ok]))
        err_r = np.abs(rew.cpu().numpy() - ref["reward"])[ok]
        if not (err_o.max() < 1e-5 and err_r.max() < 2e-5):       # diagnostics: which environments, how far from a threshold
            er_env = np.abs(rew.cpu().numpy() - ref["reward"]).max(1)
            bad = np.nonzero((er_env > 2e-5) & ok)[0]
            print(f"{tag}: reward off in envs {bad[:8]}, their fp64 margins {ref['margin'][bad[:8]]}, errors {er_env[bad[:8]]}", flush=True)
        assert err_o.max() < 1e-5 and err_r.max() < 2e-5, f"{tag}: oracle obs {err_o.max():.2e} reward {err_r.max():.2e}"
        assert np.array_equal(a.info["covered"].cpu().numpy()[ok], ref["covered"][ok]), tag
    # 6. (round 5) the raw-reward output and the host-facing step from the same state as plain steps: raw vs the oracle's
    #    uav.raw_reward, every other output bit for bit whichever way the step was asked for
    h1, h2 = make(), make()
    h1.reset(seed=c + 4); h2.reset(seed=c + 4)
    spread(h1, h2)
    rawbuf = torch.empty(1, B, N, device="cuda")
    h2.set_raw_output(rawbuf)
    for t in range(min(T, 3)):
        s2 = host(h2.get_state())
        orc.set_state(s2["ux"], s2["uy"], s2["uh"], s2["ua"], s2["tx"], s2["ty"], s2["th"], uz=s2.get("uz"), tz=s2.get("tz"))
        act = rng.randint(0, na, size=(B, N)).astype(np.int32)
        v = h1.step_host(act)
        obs, rew, _ = h2.step(torch.from_numpy(act))
        ref = orc.step(act)
        assert np.array_equal(v["obs"], obs.cpu().numpy()) and np.array_equal(v["reward"], rew.cpu().numpy()), f"{tag}: step_host t{t}"
        assert np.array_equal(v["raw"], rawbuf[0].cpu().numpy()) and np.array_equal(v["covered"], h2.info["covered"].cpu().numpy()), tag
        s3 = host(h2.get_state())
        for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th") + (("uz", "tz") if dim == 3 else ()):
            assert np.array_equal(v[k], s3[k]), f"{tag}: step_host state {k}"
        okr = ref["margin_row"] > 2.5e-4
        if okr.any():
            assert np.abs(v["raw"] - ref["raw"])[okr].max() < 1e-5, f"{tag}: raw vs oracle"
    h1.close(); h2.close()
    # 2. greedy (planar, RAW / MEAN)
    if mode != 2 and dim == 2:
        g1, g2 = make(), make()
        g1.reset(seed=c + 1); g2.reset(seed=c + 1)
        fused = g1.run_greedy(T, seed=9)
        for t in range(T):
            ga = g2.greedy_actions(seed=9)
            assert torch.equal(ga, fused["actions"][t]), f"{tag}: greedy actions t{t}"
            if t == 0:
                s2 = host(g2.get_state())
                orc.set_state(s2["ux"], s2["uy"], s2["uh"], s2["ua"], s2["tx"], s2["ty"], s2["th"])
                want, mg = greedy_actions(orc, 9, s2["step_count"], env_offset=off)
                okg = (mg["score"] > 2e-5) & (mg["angle"] > 1e-4) & (mg["dist"][:, None] > 1e-2)
                assert np.array_equal(ga.cpu().numpy()[okg], want[okg]), f"{tag}: greedy vs oracle"
            obs, rew, _ = g2.step(ga)
            assert torch.equal(obs, fused["obs"][t]) and torch.equal(rew, fused["reward"][t]), f"{tag}: run_greedy t{t}"
    # 3. actor (all modes; one action tile in 2-D, three in 3-D)
    torch.manual_seed(c)
    actor = uavtrack.ActorMLP(hidden_dim=int(rng.choice([16, 40, 128, 200])), action_dim=na)
    with torch.no_grad():
        actor.fc2.weight.mul_(5.0)
    r1, r2 = make(), make()
    r1.set_actor(actor); r2.set_actor(actor)
    o0 = r1.reset(seed=c + 2)
    obs = r2.reset(seed=c + 2).clone()
    fused = r1.run_actor(T, o0, seed=4)
    for t in range(T):
        aa, probs = r2.actor_actions(obs, seed=4, want_probs=True)
        assert torch.equal(aa, fused["actions"][t]), f"{tag}: actor actions t{t}"
        if t == 0:
            want, wp, mg = actor_actions(OracleConfig(n_envs=B, n_uav=N, m_targets=M, na=na_turn, dim=dim, nc=3 if dim == 3 else 1),
                                         obs.cpu().numpy(), actor.state_dict(), 4,
                                         r2.get_state()["step_count"].cpu().numpy(), env_offset=off)
            assert np.abs(probs.cpu().numpy() - wp).max() < 1e-5, f"{tag}: actor probs"
            oka = mg > 1e-5
            assert np.array_equal(aa.cpu().numpy()[oka], want[oka]), f"{tag}: actor vs oracle"
        o, rew, _ = r2.step(aa)
        obs = o.clone()
        assert torch.equal(obs, fused["obs"][t]) and torch.equal(rew, fused["reward"][t]), f"{tag}: run_actor t{t}"
    shard_check(c, rng, make, acts, B, off, tag, [a, b, r1, r2])


def shard_check(c, rng, make, acts, B, off, tag, to_close):
    # 5. shard == unsharded
    if B >= 3:
        lo, cnt = B // 3, B - B // 3 - 1
        full, part = make(), make(n_envs=cnt, env_offset=off + lo)
        full.reset(seed=c + 3); part.reset(seed=c + 3)
        fo = full.step_many(acts)
        po = part.step_many(acts[:, lo:lo + cnt].contiguous())
        assert torch.equal(po["obs"], fo["obs"][:, lo:lo + cnt]) and torch.equal(po["reward"], fo["reward"][:, lo:lo + cnt]), f"{tag}: shard"
    for e in to_close:
        e.close()
    print(tag, "ok", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 11)      # (another seed: another set of cases)
    for c in range(n):
        case(c, rng)
    print(f"fuzz_api ok: {n} cases")
