#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference here.

TEST INFRASTRUCTURE.  Runs only in the build container, where the reference is
mounted read-only at /root/reference; it imports src/environment.py (and what
that imports) and records inputs/outputs of Environment.reset/step as fp64 /
int arrays.  Nothing of the reference's source travels: the fixtures are data.

    PYTHONDONTWRITEBYTECODE=1 python3 oracle/gen_golden.py

Scenarios:
  g1  N5  M3   coop 0            1 seed  x 200 steps   (BASELINE configs[0])
  g2  N20 M10  coop 0   (MAAC)   8 seeds x 50 steps
  g3  N20 M10  coop .3  (MAAC-G) 8 seeds x 50 steps
  g4  N20 M10  coop .3  PMI H128 8 seeds x 50 steps   (MAAC-R)
  g4b N20 M10  coop .3  PMI H64  4 seeds x 25 steps   (MAAC-R at PMINetwork's default hidden_dim, PMINet.py:21)
  g5a N50 M25  coop 0            4 seeds x 25 steps
  g5b N50 M25  coop .3  PMI      4 seeds x 25 steps
  g6  reset-only layouts N in {5,10,20,50}
  g7  hand-placed edge cases (walls, wraps, inclusive/strict thresholds, ...)
  g9  range tests one fp32 ulp inside / outside their thresholds after the move
  g8  non-default constants (dt .5, v_max 13, h_max pi/5, dp 173.3, dc 411.7, alpha/beta/gamma .5/.3/.2, target v_max 7,
      na 9, 1500 x 1100 box, reward normalisation by config n_uav / m_targets != the environment's own): N20 M10 MAAC-G
      and MAAC-R H64, N7 M4 MAAC; 4 seeds x 25 steps each
  greedy  UAV.get_action_by_direction (uav.py:324-369, the C-METHOD baseline): best_angle per UAV on recorded states
  f3  PMINetwork.train_pmi (PMINet.py:74-100) on a recorded observation history under torch.manual_seed: the mini-batches it
      really fed to forward() (= its index draw + per-row copy + batch slicing), their outputs and the returned avg_loss
  f4  Environment.save_position / save_covered_num (environment.py:229-244) of the g1 episode: the three CSV files, as bytes
"""
import contextlib
import io
import json
import math
import os
import random
import sys

import numpy as np

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import torch  # noqa: E402
from environment import Environment  # noqa: E402  (reference)
from models.PMINet import PMINetwork  # noqa: E402  (reference)

torch.set_num_threads(1)


def make_cfg(n_uav, m_targets, coop, x_max=2000, y_max=2000, na=12):
    # identical environment/uav/target blocks of configs/*.yaml
    return {
        "environment": {"n_uav": n_uav, "m_targets": m_targets, "x_max": x_max, "y_max": y_max, "na": na},
        "uav": {"dt": 1, "v_max": 20, "h_max": 6, "dc": 500, "dp": 200, "alpha": 0.6, "beta": 0.2, "gamma": 0.2},
        "target": {"v_max": 5, "h_max": 6},
        "cooperative": coop,
    }


def snap(env):
    u, t = env.uav_list, env.target_list
    return dict(ux=[a.x for a in u], uy=[a.y for a in u], uh=[a.h for a in u], ua=[a.a for a in u],
                tx=[a.x for a in t], ty=[a.y for a in t], th=[a.h for a in t])


def make_pmi(hidden=128, seed=42):
    torch.manual_seed(seed)
    pmi = PMINetwork(hidden_dim=hidden)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for bn in (pmi.bn_comm, pmi.bn_obs, pmi.bn_boundary_state, pmi.bn1):
            bn.running_mean.copy_(torch.randn(hidden, generator=g) * 0.3)
            bn.running_var.copy_(torch.rand(hidden, generator=g) * 1.5 + 0.5)
            bn.weight.copy_(torch.rand(hidden, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(hidden, generator=g) * 0.2)
    pmi.eval()
    return pmi


def run_episode(cfg, pmi, steps, seed=None, init=None, actions=None, step_cfg=None):
    """Returns dict of stacked arrays: states [T+1,...], actions [T,N], outputs [T,...].
    step_cfg: the config dict handed to Environment.step when it differs from the one reset() saw (the reward
    normalisation reads config['environment']['n_uav' / 'm_targets'] per step, environment.py:207-210)."""
    e = cfg["environment"]
    env = Environment(n_uav=e["n_uav"], m_targets=e["m_targets"], x_max=e["x_max"], y_max=e["y_max"], na=e["na"])
    if seed is not None:
        random.seed(seed)
    env.reset(config=cfg)
    if init is not None:
        for i, (x, y, h, a) in enumerate(init["uav"]):
            u = env.uav_list[i]
            u.x, u.y, u.h, u.a = float(x), float(y), float(h), int(a)
        for k, (x, y, h) in enumerate(init["target"]):
            t = env.target_list[k]
            t.x, t.y, t.h = float(x), float(y), float(h)
    with contextlib.redirect_stdout(io.StringIO()):
        obs0 = np.array(env.get_states(), dtype=np.float64)
    states = [snap(env)]
    acts, obs, rew, terms, raw, cov = [], [], [], [], [], []
    sink = io.StringIO()
    for t in range(steps):
        if actions is not None:
            a = list(actions[t])
        else:
            a = [random.randint(0, e["na"] - 1) for _ in range(e["n_uav"])]
        with contextlib.redirect_stdout(sink):
            nxt, r, c = env.step(step_cfg or cfg, pmi, a)
        acts.append(a)
        obs.append(np.array(nxt, dtype=np.float64))
        rew.append(np.array(r["rewards"], dtype=np.float64))
        terms.append(np.array([r["target_tracking_reward"], r["boundary_punishment"],
                               r["duplicate_tracking_punishment"]], dtype=np.float64))
        raw.append(np.array([u.raw_reward for u in env.uav_list], dtype=np.float64))
        cov.append(c)
        states.append(snap(env))
    out = {k: np.array([s[k] for s in states], dtype=(np.int64 if k == "ua" else np.float64))
           for k in states[0]}
    out.update(actions=np.array(acts, dtype=np.int64).reshape(steps, e["n_uav"]),
               obs=np.array(obs), reward=np.array(rew), terms=np.array(terms), raw=np.array(raw),
               covered=np.array(cov, dtype=np.int64), obs0=obs0,
               overstep_prints=np.int64(sink.getvalue().count("overstep")))
    return out


def stack_eps(eps):
    return {k: np.stack([e[k] for e in eps]) for k in eps[0]}


def save(name, arrays, meta):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def scenario(name, n, m, coop, pmi, seeds, steps):
    cfg = make_cfg(n, m, coop)
    eps = [run_episode(cfg, pmi, steps, seed=s) for s in seeds]
    save(name, stack_eps(eps), dict(n_uav=n, m_targets=m, cooperative=coop, pmi=pmi is not None,
                                    seeds=list(seeds), steps=steps, cfg=cfg))


def gen_reset():
    arrays, meta = {}, {}
    for n, m in ((5, 3), (10, 10), (20, 10), (50, 25)):
        cfg = make_cfg(n, m, 0)
        ep = run_episode(cfg, None, 0, seed=42)
        for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th", "obs0"):
            arrays[f"n{n}_{k}"] = ep[k][0] if k != "obs0" else ep[k]
        meta[f"n{n}"] = dict(n_uav=n, m_targets=m)
    save("g6_reset", arrays, meta)


def gen_edges(pmi):
    """Hand-placed single/double-step cases; each gets its own key prefix."""
    P = math.pi
    cases = []

    def add(name, uav, target, actions, coop=0.0, use_pmi=False):
        cases.append(dict(name=name, uav=uav, target=target, actions=actions, coop=coop, use_pmi=use_pmi))

    far_t = [(1000.0, 1900.0, 0.3)]
    # targets crossing each wall, a corner (y has priority), h>0 vs h<=0 at x walls
    add("target_walls",
        uav=[(300.0, 300.0, 0.0, 3)],
        target=[(1998.0, 1000.0, 0.1), (1998.0, 1000.0, -0.1), (2.0, 1000.0, P - 0.1), (2.0, 1000.0, -P + 0.1),
                (1000.0, 1998.0, P / 2), (1000.0, 2.0, -P / 2), (1998.0, 1998.0, P / 4), (1998.0, 2.0, -P / 4),
                (2.0, 2.0, -3 * P / 4), (1999.0, 500.0, 0.0)],
        actions=[[5], [6], [6]])
    # heading wrap both ways
    add("heading_wrap",
        uav=[(1000.0, 1000.0, P - 0.05, 0), (1200.0, 1000.0, -P + 0.05, 0), (800.0, 900.0, P - 0.3, 4)],
        target=far_t, actions=[[11, 0, 11], [11, 0, 11], [0, 11, 11]])
    # outside the box (-0.5 -> bp -1) and d_b exactly dp (not < dp -> 0)
    add("boundary",
        uav=[(-30.0, 1000.0, 0.0, 5), (180.0, 1000.0, 0.0, 5), (1000.0, 1995.0, P / 2, 5), (1000.0, 150.0, 0.0, 5)],
        target=far_t, actions=[[5, 5, 6, 6]])
    # inclusive thresholds: target exactly dp (obs/track yes, coverage no); peer exactly dc; 2dp; dp
    add("thresholds",
        uav=[(480.0, 1000.0, 0.0, 2), (1000.0, 1000.0, 0.0, 7), (880.0, 400.0, 0.0, 1), (480.0, 400.0, 0.0, 9),
             (680.0, 1600.0, 0.0, 3), (480.0, 1600.0, 0.0, 3)],
        target=[(695.0, 1000.0, 0.0), (200.0, 1000.0, 0.0)],
        actions=[[5, 6, 5, 6, 5, 6]], coop=0.3)
    add("thresholds_pmi",
        uav=[(480.0, 1000.0, 0.0, 2), (1000.0, 1000.0, 0.0, 7), (880.0, 400.0, 0.0, 1), (480.0, 400.0, 0.0, 9),
             (680.0, 1600.0, 0.0, 3), (480.0, 1600.0, 0.0, 3)],
        target=[(695.0, 1000.0, 0.0), (200.0, 1000.0, 0.0)],
        actions=[[5, 6, 5, 6, 5, 6]], coop=0.3, use_pmi=True)
    # uav.py:165/179 weight min(d,1) < 1: UAV ends within 1 m of its normalised offsets
    add("near_origin_weight",
        uav=[(-19.7, 0.2, 0.0, 4), (100.0, 50.0, 1.0, 2), (-19.9, 0.6, 0.0, 8)],
        target=[(50.0, 60.0, 0.5), (120.0, -10.0, 2.0)],
        actions=[[5, 6, 7]], coop=0.3)
    # nothing observed, no neighbours: MAAC-G gives 0 (uav.py:308-309), MAAC-R gives (1-a)*raw (uav.py:290)
    add("isolated_mean",
        uav=[(100.0, 100.0, 0.3, 0), (1900.0, 1900.0, -2.0, 11), (100.0, 1900.0, 1.0, 4)],
        target=[(1000.0, 1000.0, 0.0)], actions=[[1, 2, 3]], coop=0.3)
    add("isolated_pmi",
        uav=[(100.0, 100.0, 0.3, 0), (1900.0, 1900.0, -2.0, 11), (100.0, 1900.0, 1.0, 4)],
        target=[(1000.0, 1000.0, 0.0)], actions=[[1, 2, 3]], coop=0.3, use_pmi=True)
    # all UAVs coincident: max duplicate punishment, j<i sees d=0, j>i sees d=20
    add("coincident",
        uav=[(1000.0, 1000.0, 0.7, 3)] * 5,
        target=[(1010.0, 1010.0, 0.0), (1500.0, 1500.0, 1.0)],
        actions=[[6, 6, 6, 6, 6], [0, 11, 5, 6, 3]], coop=0.3)
    add("coincident_pmi",
        uav=[(1000.0, 1000.0, 0.7, 3)] * 5,
        target=[(1010.0, 1010.0, 0.0), (1500.0, 1500.0, 1.0)],
        actions=[[6, 6, 6, 6, 6], [0, 11, 5, 6, 3]], coop=0.3, use_pmi=True)
    # dense cluster, many targets in range: tracking sum, coverage, softmax over many neighbours
    rng = np.random.RandomState(7)
    n, m = 12, 8
    add("dense_cluster_pmi",
        uav=[(float(900 + rng.rand() * 250), float(900 + rng.rand() * 250), float(rng.uniform(-P, P)),
              int(rng.randint(0, 12))) for _ in range(n)],
        target=[(float(850 + rng.rand() * 350), float(850 + rng.rand() * 350), float(rng.uniform(-P, P)))
                for _ in range(m)],
        actions=[[int(a) for a in rng.randint(0, 12, size=n)] for _ in range(3)], coop=0.3, use_pmi=True)

    run_cases("g7_edges", cases, pmi)


def run_cases(name, cases, pmi):
    arrays, meta = {}, {"cases": []}
    for c in cases:
        n, m = len(c["uav"]), len(c["target"])
        cfg = make_cfg(n, m, c["coop"])
        ep = run_episode(cfg, pmi if c["use_pmi"] else None, len(c["actions"]), seed=1,
                         init=dict(uav=c["uav"], target=c["target"]), actions=c["actions"])
        for k, v in ep.items():
            arrays[f"{c['name']}__{k}"] = v
        meta["cases"].append(dict(name=c["name"], n_uav=n, m_targets=m, cooperative=c["coop"],
                                  pmi=c["use_pmi"], steps=len(c["actions"])))
    save(name, arrays, meta)


def gen_ulp_edges():
    """g9: range tests ONE fp32 ULP either side of their threshold AFTER the move (g7's threshold cases sit exactly on
    the thresholds, all representable).  Everything flies along y = 1000 with heading 0, so a move adds exactly 20 m
    (UAV) / 5 m (target) to x and nothing to y: the post-move coordinates are fp32 numbers in the reference's fp64
    arithmetic and in the kernel's fp32 alike, and so are their differences -- the only thing under test is the
    comparison itself.  UAV 0 ends at x = 56.  `inside`: target 0 ends one ulp(256) short of 256 (d = dp - 2^-16:
    observed, tracked AND covered), UAV 1 one ulp(512) short of 556 (d = dc - 2^-14: communicates), UAV 2 one ulp(256)
    short of 456 (d = 2 dp - 2^-15: duplicate punishment), UAV 3 one ulp short of 256 at y = 1000 (d = dp - 2^-16 to UAV 0:
    a cooperative neighbour).  `outside`: the same four one ulp beyond.  MAAC-G so that the neighbour test shows."""
    u = lambda e: 2.0 ** e                     # noqa: E731
    cases = []
    for name, sgn in (("ulp_inside", -1.0), ("ulp_outside", +1.0)):
        # (below a power of two the spacing halves: one ulp short of 256 is 256 - 2^-16, one ulp beyond is 256 + 2^-15)
        t0 = 256.0 + (sgn * u(-16) if sgn < 0 else u(-15))
        u1 = 556.0 + sgn * u(-14)
        u2 = 456.0 + sgn * u(-15)
        u3 = 256.0 + (sgn * u(-16) if sgn < 0 else u(-15))
        cases.append(dict(name=name,
                          uav=[(56.0 - 20.0, 1000.0, 0.0, 3), (u1 - 20.0, 1000.0, 0.0, 7), (u2 - 20.0, 1000.0, 0.0, 1),
                               (u3 - 20.0, 1000.0, 0.0, 9)],
                          target=[(t0 - 5.0, 1000.0, 0.0), (1500.0, 300.0, 0.0)],
                          actions=[[5, 6, 2, 9]], coop=0.3, use_pmi=False))
    for c in cases:       # every coordinate must be an fp32 number before and after the move
        for (x, y, h, a), v in zip(c["uav"], [20.0] * 4):
            assert float(np.float32(x)) == x and float(np.float32(x + v)) == x + v, (c["name"], x)
        assert float(np.float32(c["target"][0][0])) == c["target"][0][0]
        assert float(np.float32(c["target"][0][0] + 5.0)) == c["target"][0][0] + 5.0
    run_cases("g9_ulp_edges", cases, None)


def gen_actor():
    """FnnPolicyNet.forward (actor_critic.py:85-98) on recorded observations: weights + obs -> probs (fp32)."""
    from models.actor_critic import FnnPolicyNet  # noqa: E402  (reference)
    torch.manual_seed(7)
    net = FnnPolicyNet(12, 128, 12)          # configs/MAAC.yaml:34 hidden_dim 128; na 12
    with torch.no_grad():                    # default init gives a near-uniform policy; sharpen it (weights are data)
        net.fc2.weight.mul_(6.0)
        net.fc2.bias.uniform_(-1.0, 1.0)
    g2 = np.load(os.path.join(OUT, "g2_n20m10_raw.npz"))
    obs = g2["obs"].reshape(-1, 12).astype(np.float32)            # what get_local_state returned in G2
    rng = np.random.default_rng(7)
    obs = np.concatenate([obs, rng.normal(0.0, 2.0, size=(256, 12)).astype(np.float32)])
    with torch.no_grad():
        probs = net(torch.from_numpy(obs)).numpy()
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in net.state_dict().items()}
    save("actor_h128", dict(obs=obs, probs=probs.astype(np.float32), **{k.replace(".", "__"): v for k, v in sd.items()}),
         dict(hidden=128, na=12, torch_seed=7, source="FnnPolicyNet(12,128,12), fc2.weight x6, fc2.bias U(-1,1)"))


def gen_greedy():
    """UAV.get_action_by_direction (uav.py:324-369).  Upstream it ends in a call to an undefined
    find_closest_a_idx (uav.py:368), so the method cannot return; what it computes up to that call can be
    recorded: a recorder is attached to the imported class under that name (it receives best_angle) and
    random.random is pinned to 0.99 for the call, which takes the non-epsilon (uav.py:338) and the
    non-keep-straight (uav.py:365) branch -- the deterministic target-scoring rule of uav.py:341-362.
    States: free-running N20 M10 episodes in the reference box, and N10 M10 in a 600 m box (targets crowded
    by several UAVs within dc: the 0.8 penalties decide)."""
    from agent.uav import UAV  # noqa: E402  (reference)
    seen = []
    UAV.find_closest_a_idx = lambda self, angle: (seen.append(float(angle)), 0)[1]
    arrays, meta = {}, {}
    for tag, n, m, box, seeds, every in (("n20m10", 20, 10, 2000, (42, 43, 44, 45), 10), ("n10m10_box600", 10, 10, 600, (7, 8, 9, 10), 6)):
        cfg = make_cfg(n, m, 0, x_max=box, y_max=box)
        states, angles = [], []
        for sd in seeds:
            env = Environment(n_uav=n, m_targets=m, x_max=box, y_max=box, na=12)
            random.seed(sd)
            env.reset(config=cfg)
            for t in range(41):
                if t % every == 0:
                    real = random.random
                    random.random = lambda: 0.99
                    try:
                        row = []
                        for u in env.uav_list:
                            del seen[:]
                            u.get_action_by_direction(env.target_list, env.uav_list)
                            assert len(seen) == 1
                            row.append(seen[0])
                    finally:
                        random.random = real
                    states.append(snap(env))
                    angles.append(row)
                with contextlib.redirect_stdout(io.StringIO()):
                    env.step(cfg, None, [random.randint(0, 11) for _ in range(n)])
        for k in states[0]:
            arrays[f"{tag}__{k}"] = np.array([st[k] for st in states], dtype=(np.int64 if k == "ua" else np.float64))
        arrays[f"{tag}__best_angle"] = np.array(angles, dtype=np.float64)
        meta[tag] = dict(n_uav=n, m_targets=m, box=box, seeds=list(seeds), states=len(states), cfg=cfg)
    del UAV.find_closest_a_idx
    save("greedy_ref", arrays, meta)


def gen_h64():
    """MAAC-R with PMINetwork at its class-default width (PMINet.py:21, hidden_dim=64): weights + a short scenario."""
    pmi = make_pmi(64, 43)
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in pmi.state_dict().items()
          if "num_batches_tracked" not in k}
    save("pmi_h64", sd, dict(hidden=64, bn_eps=1e-5, torch_seed=43))
    scenario("g4b_n20m10_pmi_h64", 20, 10, 0.3, pmi, [52, 53, 54, 55], 25)


def gen_nondefault():
    """Every constant of the per-step path away from configs/*.yaml, and the reward normalisation keyed by a config
    whose n_uav / m_targets differ from the Environment's own (environment.py:207-210 read the dict, not the object)."""
    pmi = make_pmi(64, 43)
    arrays, meta = {}, {"cases": []}
    for tag, n, m, coop, net in (("n20m10_mean", 20, 10, 0.3, None), ("n20m10_pmi_h64", 20, 10, 0.3, pmi), ("n7m4_raw", 7, 4, 0.0, None)):
        cfg = {
            "environment": {"n_uav": n, "m_targets": m, "x_max": 1500, "y_max": 1100, "na": 9},
            "uav": {"dt": 0.5, "v_max": 13, "h_max": 5, "dc": 411.7, "dp": 173.3, "alpha": 0.5, "beta": 0.3, "gamma": 0.2},
            "target": {"v_max": 7, "h_max": 6},
            "cooperative": coop,
        }
        step_cfg = json.loads(json.dumps(cfg))
        step_cfg["environment"]["n_uav"] = n + 4          # the clip's N and M (environment.py:208,210)
        step_cfg["environment"]["m_targets"] = max(1, m - 3)
        eps = [run_episode(cfg, net, 25, seed=s, step_cfg=step_cfg) for s in (61, 62, 63, 64)]
        for k, v in stack_eps(eps).items():
            arrays[f"{tag}__{k}"] = v
        meta["cases"].append(dict(name=tag, n_uav=n, m_targets=m, cooperative=coop, pmi=net is not None, seeds=[61, 62, 63, 64],
                                  steps=25, cfg=cfg, norm_n_uav=n + 4, norm_m_targets=max(1, m - 3)))
    save("g8_nondefault", arrays, meta)


def gen_pmi_train():
    """f3.  PMINetwork.train_pmi (PMINet.py:74-100) draws torch.randint triples from torch's global generator, copies the
    rows one by one and walks b2_size // batch_size mini-batches.  What it selected is only visible as the arguments of its
    own forward() calls, so a recorder is put in front of the instance's forward: inputs and outputs of every call."""
    n_uav, steps, b2, bs = 20, 50, 300, 64
    ep = run_episode(make_cfg(n_uav, 10, 0), None, steps, seed=42)            # the g2 episode of seed 42
    train = torch.tensor(ep["obs"].reshape(steps * n_uav, 12), dtype=torch.float32)   # train.py:183-184 order: step-major, then UAV
    torch.manual_seed(11)
    pmi = PMINetwork(hidden_dim=64, b2_size=b2)
    calls = []
    inner = pmi.forward

    def recorder(x):
        y = inner(x)
        calls.append((x.detach().clone().numpy(), y.detach().clone().numpy()))
        return y
    pmi.forward = recorder
    torch.manual_seed(123)                                                    # the seed the test replays
    avg_loss = pmi.train_pmi({"pmi": {"batch_size": bs}}, train, n_uav)
    nb = b2 // bs
    assert len(calls) == 2 * nb
    save("f3_pmi_train", dict(train_data=train.numpy(),
                              in_1_2=np.stack([calls[2 * i][0] for i in range(nb)]), in_1_3=np.stack([calls[2 * i + 1][0] for i in range(nb)]),
                              out_1_2=np.stack([calls[2 * i][1] for i in range(nb)]), out_1_3=np.stack([calls[2 * i + 1][1] for i in range(nb)]),
                              avg_loss=np.float64(avg_loss)),
         dict(n_uav=n_uav, steps=steps, b2_size=b2, batch_size=bs, torch_seed=123, hidden=64,
              source="PMINetwork.train_pmi, PMINet.py:74-100; obs history = the seed-42 episode of g2"))


def gen_export():
    """f4.  The g1 episode (N5 M3, random.seed(42), 200 steps) through the reference's own writers."""
    import tempfile
    cfg = make_cfg(5, 3, 0)
    env = Environment(n_uav=5, m_targets=3, x_max=2000, y_max=2000, na=12)
    random.seed(42)
    env.reset(config=cfg)
    sink = io.StringIO()
    for t in range(200):
        a = [random.randint(0, 11) for _ in range(5)]
        with contextlib.redirect_stdout(sink):
            env.step(cfg, None, a)
    with tempfile.TemporaryDirectory() as d:
        for sub in ("u_xy", "t_xy", "covered_target_num"):
            os.makedirs(os.path.join(d, sub))
        env.save_position(d, 7)
        env.save_covered_num(d, 7)
        files = {k: np.frombuffer(open(os.path.join(d, k, f"{k}7.csv"), "rb").read(), dtype=np.uint8)
                 for k in ("u_xy", "t_xy", "covered_target_num")}
    save("f4_export", files, dict(episode="g1_n5m3_raw (random.seed(42), 200 steps)", epoch_i=7,
                                  source="Environment.save_position / save_covered_num, environment.py:229-244"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--only-f3f4":
        gen_pmi_train()
        gen_export()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-nondefault":
        gen_nondefault()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-actor":
        gen_actor()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-greedy":
        gen_greedy()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-ulp":
        gen_ulp_edges()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "--only-h64":
        gen_h64()
        return
    pmi = make_pmi(128, 42)
    sd = {k: v.detach().numpy().astype(np.float32) for k, v in pmi.state_dict().items()
          if "num_batches_tracked" not in k}
    save("pmi_h128", sd, dict(hidden=128, bn_eps=1e-5, torch_seed=42))
    scenario("g1_n5m3_raw", 5, 3, 0, None, [42], 200)
    s8 = [42, 43, 44, 45, 46, 47, 48, 49]
    scenario("g2_n20m10_raw", 20, 10, 0, None, s8, 50)
    scenario("g3_n20m10_mean", 20, 10, 0.3, None, s8, 50)
    scenario("g4_n20m10_pmi", 20, 10, 0.3, pmi, s8, 50)
    scenario("g5a_n50m25_raw", 50, 25, 0, None, [42, 43, 44, 45], 25)
    scenario("g5b_n50m25_pmi", 50, 25, 0.3, pmi, [42, 43, 44, 45], 25)
    gen_reset()
    gen_edges(pmi)
    gen_actor()
    gen_greedy()
    gen_h64()       # (nothing above depends on what it does to the global RNGs)
    gen_nondefault()
    gen_pmi_train()
    gen_export()
    gen_ulp_edges()


if __name__ == "__main__":
    main()
