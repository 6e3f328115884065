#!/usr/bin/env python3
"""Long teacher-forced comparison HIP vs oracle (GPU box; run by hand: `python tests/soak.py`, not collected by pytest): many envs,
many steps, several boxes, so that rare paths (wall mirrors, heading wraps, near-origin weights,
UAVs outside the box, crowded neighbourhoods) all occur."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # tests/ -> repo root
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
import torch
import uavtrack
from oracle import OracleConfig, OracleEnv

MARGIN = 2.5e-4      # fp64 |d - threshold| below which a result is set aside (tests/test_hip_parity.py)


def run(B, N, M, coop, box, steps, seed, dim=2, z_max=300.0, spread_z=False):
    """Per ROW (oracle margin_row: the UAV's own range tests) for the observation row, the terms and the MAAC reward; per
    ENVIRONMENT (margin: every test) for the coverage count and the cooperative rewards -- as in the suite."""
    kw = dict(n_envs=B, n_uav=N, m_targets=M, cooperative=coop, x_max=box, y_max=box, dim=dim, nc=3 if dim == 3 else 1, z_max=z_max)
    env = uavtrack.BatchedUavEnv(uavtrack.EnvConfig(**kw)); env.reset(seed=seed)
    orc = OracleEnv(OracleConfig(**kw), n_threads=16)
    rng = np.random.RandomState(seed)
    if dim == 3 and spread_z:       # a reset puts every UAV at z_max / 2: spread the swarm and the targets over the whole altitude band
        st = {k: v for k, v in env.get_state().items()}
        st["uz"] = torch.from_numpy(rng.uniform(0.0, z_max, size=(B, N)).astype(np.float32)).cuda()
        st["tz"] = torch.from_numpy(rng.uniform(0.0, z_max, size=(B, M)).astype(np.float32)).cuda()
        env.set_state(**st)
    worst = dict(obs=0.0, rew=0.0, terms=0.0); bad_cov = 0
    rows_aside = rows = envs_aside = envs = 0
    cen = dict(covered=0, obs=0, reward=0)          # of what is set aside: how much REALLY differs from the oracle
    na = 12 * (3 if dim == 3 else 1)
    for t in range(steps):
        st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
        orc.set_state(st["ux"], st["uy"], st["uh"], st["ua"], st["tx"], st["ty"], st["th"], uz=st.get("uz"), tz=st.get("tz"))
        act = rng.randint(0, na, size=(B, N)).astype(np.int32)
        obs, rew, _ = env.step(torch.from_numpy(act))
        ref = orc.step(act)
        # (the margin is two fp32 ulps of a pose: 2.5e-4 m in the reference's 2000 m box, more in a larger one)
        mg = MARGIN * max(1.0, box / 2000.0)
        ok, okr = ref["margin"] > mg, ref["margin_row"] > mg
        rok = okr if coop == 0 else np.broadcast_to(ok[:, None], okr.shape)      # the reward's mask: its own row (MAAC) or the environment
        rows_aside += int((~okr).sum()); rows += okr.size; envs_aside += int((~ok).sum()); envs += B
        o = obs.cpu().numpy(); r = rew.cpu().numpy(); tm = env.info["terms"].cpu().numpy(); cv = env.info["covered"].cpu().numpy()
        eo = (np.abs(o - ref["obs"]) / (1.0 + np.abs(ref["obs"]))).max(-1)          # [B, N]
        er = np.abs(r - ref["reward"])
        cen["covered"] += int(((cv != ref["covered"]) & ~ok).sum())
        cen["obs"] += int(((eo > 1e-5) & ~okr).sum())
        cen["reward"] += int(((er > 1e-5) & ~rok).sum())
        # near the origin the uav.py:165 weight 1/min(d,1) makes observation entries O(10..1000): mixed abs/rel, and a UAV whose
        # post-move pose lies within 2.5 m of the origin (the kernel's own test for the weighted path) is held to the documented
        # near-origin bound 2e-4, every other row to 1e-5
        st2 = {k: v.cpu().numpy() for k, v in env.get_state().items()}
        near = (np.abs(st2["ux"]) < 2.5) & (np.abs(st2["uy"]) < 2.5)
        if (okr & near).any():
            worst["obs_near_origin"] = max(worst.get("obs_near_origin", 0.0), float(eo[okr & near].max()))
            assert worst["obs_near_origin"] < 2e-4
        if (okr & ~near).any():
            worst["obs"] = max(worst["obs"], float(eo[okr & ~near].max()))
        if rok.any():
            worst["rew"] = max(worst["rew"], float(er[rok].max()))
        if okr.any():
            worst["terms"] = max(worst["terms"], float(np.abs(tm - ref["terms"])[:, okr].max()))
        bad_cov += int((cv != ref["covered"])[ok].sum())
    print(f"B{B} N{N} M{M} coop{coop} box{box} dim{dim}{' z-spread ' + str(z_max) if spread_z else ''} steps{steps}: max|obs|={worst['obs']:.2e} (near origin {worst.get('obs_near_origin', 0.0):.2e}) max|rew|={worst['rew']:.2e} "
          f"max|terms|={worst['terms']:.2e} covered mismatches={bad_cov}; set aside: rows {rows_aside}/{rows}, envs {envs_aside}/{envs}; of those "
          f"really different: covered {cen['covered']}, obs rows {cen['obs']}, reward rows {cen['reward']}", flush=True)
    CENSUS.append((rows_aside, rows, envs_aside, envs, cen["covered"], cen["obs"], cen["reward"]))
    assert worst["obs"] < 1e-5 and worst["rew"] < 1e-5 and worst["terms"] < 1e-5 and bad_cov == 0

CENSUS = []


def fuzz(n_cases, seed):
    """Random shapes / boxes / modes through the same comparison: odd and even N and M, one UAV, one target,
    more than 32 and 64 targets (multi-word coverage masks), N past a wavefront, tiny and huge boxes, 2-D and 3-D."""
    rng = np.random.RandomState(seed)
    for c in range(n_cases):
        N = int(rng.choice([1, 2, 3, 5, 7, 8, 11, 16, 19, 20, 21, 31, 32, 33, 47, 50, 63, 64, 65, 96, 127]))
        M = int(rng.choice([1, 2, 3, 4, 9, 10, 17, 25, 31, 32, 33, 40, 63, 64, 65, 70]))
        B = int(rng.choice([1, 2, 3, 17, 64, 100, 257]))
        if N * B > 12000:
            B = max(1, 12000 // N)
        coop = float(rng.choice([0.0, 0.3, 0.9]))
        box = float(rng.choice([50.0, 300.0, 1000.0, 2000.0, 10000.0]))
        dim = int(rng.choice([2, 2, 3]))
        run(B, N, M, coop, box, int(rng.choice([3, 8, 20])), 1000 + c, dim=dim, z_max=float(rng.choice([100.0, 300.0, 1000.0])),
            spread_z=bool(rng.randint(0, 2)))


t0 = time.time()
if "--fuzz" in sys.argv:
    fuzz(int(sys.argv[sys.argv.index("--fuzz") + 1]), int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 7)
    print(f"fuzz ok in {time.time()-t0:.0f} s")
    sys.exit(0)
if "--quick" not in sys.argv:
    run(4096, 20, 10, 0.0, 2000.0, 200, 1)
    run(4096, 20, 10, 0.3, 2000.0, 100, 2)
run(2048, 20, 10, 0.3, 300.0, 150, 3)     # tiny box: everything in range, UAVs leave the box, many reflections
run(1024, 50, 25, 0.3, 2000.0, 60, 4)
run(1024, 50, 25, 0.0, 2000.0, 40, 5, dim=3)
run(1024, 20, 10, 0.3, 400.0, 60, 7, dim=3, z_max=600.0, spread_z=True)     # 3-D, altitudes over the whole band: planar and spatial ranges differ
run(2048, 7, 4, 0.3, 100.0, 300, 6)       # generic kernel, box smaller than a step: origin-weight path, outside-box UAVs
tot = np.array(CENSUS).sum(0)
print(f"census: {tot[1]} UAV-steps compared, {tot[0]} set aside on a knife edge ({100.0 * tot[0] / tot[1]:.4f} %); {tot[3]} env-steps, "
      f"{tot[2]} set aside ({100.0 * tot[2] / tot[3]:.3f} %); of what is set aside the device really differs from the fp64 oracle in: "
      f"covered count {tot[4]} env-steps, observation {tot[5]} rows, reward {tot[6]} rows")
print(f"soak ok in {time.time()-t0:.0f} s")
