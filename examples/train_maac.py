#!/usr/bin/env python3
"""End-to-end MAAC training loop on the batched environment (example; the learner is plain PyTorch).

What the reference does per episode (src/train.py:142-196, 199-290): 200 steps x N batch-1 actor calls +
Environment.step, push the N*200 transitions into the replay buffer, sample, one ActorCritic.update
(src/models/actor_critic.py:150-178: TD(0) target r + gamma*V(s'), actor loss -log pi(a|s) * delta, critic MSE).
Here one iteration = B episodes at once: the whole rollout (actor + environment, B x N x T agent-steps) is ONE
launch of the library (uavtrack_run_actor), its [T,B,N] outputs go straight into a device replay ring, the
update is the same rule on a sampled batch, and the new actor weights are re-uploaded (sync_actor).

    python examples/train_maac.py --envs 1024 --iters 40
    python examples/train_maac.py --method maac-r --envs 1024 --iters 40     # reciprocal (PMI) reward, PMI net trained too

--method maac-r is the paper's method (configs/MAAC-R.yaml): the reward of every step is mixed in-kernel with the
neighbours' rewards, weighted by the PMI network's scores (uav.py:262-291); that network is trained alongside on
(timestep, uav-pair) samples of the rollout's observations (PMINet.py:74-100, here uavtrack.sample_pmi_pairs +
pmi_contrastive_loss on the device history) and its BatchNorm-folded weights are re-uploaded every iteration.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]

import torch  # noqa: E402
import uavtrack  # noqa: E402


class ValueNet(torch.nn.Module):
    """Same shape as the reference's critic FnnValueNet (actor_critic.py:101-112): Linear-ReLU-Linear -> scalar."""

    def __init__(self, state_dim=12, hidden_dim=128):
        super().__init__()
        self.fc1 = torch.nn.Linear(state_dim, hidden_dim)
        self.fc2 = torch.nn.Linear(hidden_dim, 1)

    def forward(self, x):
        return self.fc2(torch.relu(self.fc1(x))).squeeze(-1)


def update(actor, critic, opt_a, opt_c, batch, gamma):
    """One ActorCritic.update step (actor_critic.py:150-178) on a batch of transitions."""
    s, a, r, s2 = batch["states"], batch["actions"].long().unsqueeze(1), batch["rewards"], batch["next_states"]
    td_target = r + gamma * critic(s2)
    td_delta = td_target - critic(s)
    log_probs = torch.log(actor(s).gather(1, a).squeeze(1).clamp_min(1e-12))
    actor_loss = torch.mean(-log_probs * td_delta.detach())
    critic_loss = torch.nn.functional.mse_loss(critic(s), td_target.detach())
    opt_a.zero_grad(); opt_c.zero_grad()
    actor_loss.backward(); critic_loss.backward()
    opt_a.step(); opt_c.step()
    return float(actor_loss.detach()), float(critic_loss.detach())


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--n-uav", type=int, default=10)        # configs/MAAC.yaml: 10 UAVs, 10 targets
    ap.add_argument("--m-targets", type=int, default=10)
    ap.add_argument("--steps", type=int, default=200)       # main.py:128 num_steps
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--updates", type=int, default=8, help="learner updates per iteration")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--hidden", type=int, default=128)      # configs/MAAC.yaml:34
    ap.add_argument("--gamma", type=float, default=0.95)
    ap.add_argument("--actor-lr", type=float, default=1e-4)
    ap.add_argument("--critic-lr", type=float, default=5e-4)
    ap.add_argument("--method", choices=["maac", "maac-g", "maac-r"], default="maac")
    ap.add_argument("--cooperative", type=float, default=None, help="default: 0 for maac, 0.3 for maac-g / maac-r")
    ap.add_argument("--pmi-hidden", type=int, default=128)   # configs/MAAC-R.yaml:39
    ap.add_argument("--pmi-b2", type=int, default=3000)      # PMINetwork b2_size (PMINet.py:21)
    ap.add_argument("--pmi-batch", type=int, default=500)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args(argv)

    dev = "cuda:0"
    torch.manual_seed(args.seed)
    coop = args.cooperative if args.cooperative is not None else (0.0 if args.method == "maac" else 0.3)
    mode = {"maac": uavtrack.RewardMode.RAW, "maac-g": uavtrack.RewardMode.MEAN, "maac-r": uavtrack.RewardMode.PMI}[args.method]
    cfg = uavtrack.EnvConfig(n_envs=args.envs, n_uav=args.n_uav, m_targets=args.m_targets, cooperative=coop,
                             reward_mode=mode, horizon=args.steps)
    env = uavtrack.BatchedUavEnv(cfg, dev)
    pmi = opt_p = None
    if args.method == "maac-r":
        pmi = uavtrack.make_pmi_net(args.pmi_hidden).to(dev)
        opt_p = torch.optim.Adam(pmi.parameters(), lr=1e-3)          # PMINet.py:39
        env.set_pmi(pmi.state_dict())
    actor = uavtrack.ActorMLP(hidden_dim=args.hidden, action_dim=cfg.na_total).to(dev)
    critic = ValueNet(hidden_dim=args.hidden).to(dev)
    opt_a = torch.optim.Adam(actor.parameters(), lr=args.actor_lr)
    opt_c = torch.optim.Adam(critic.parameters(), lr=args.critic_lr)
    rollout = uavtrack.BatchedRollout(env, actor, device_actor=True, seed=args.seed)
    per_iter = args.envs * args.n_uav * args.steps
    replay = uavtrack.DeviceReplayBuffer(capacity=2 * per_iter, device=dev)
    history = []
    out = None
    for it in range(args.iters):
        t0 = time.perf_counter()
        rollout.seed = args.seed + it
        rollout.reset(seed=1000 + it)
        obs_in = rollout.obs.clone()
        res = rollout.run_fused(args.steps, out=out)                  # B episodes, one launch
        out = {k: v for k, v in res.items() if k != "ep_sums"}        # reuse the output buffers next time
        replay.add(uavtrack.transitions_from_rollout(obs_in, res))
        torch.cuda.synchronize()
        t_roll = time.perf_counter() - t0
        for _ in range(args.updates):
            la, lc = update(actor, critic, opt_a, opt_c, replay.sample(args.batch), args.gamma)
        lp = float("nan")
        if pmi is not None:                                           # PMINetwork.train_pmi on this rollout's observations
            pmi.train()
            sel, _, _ = uavtrack.sample_pmi_pairs(res["obs"], args.n_uav, args.pmi_b2)
            for x12, x13 in uavtrack.pmi_batches(sel, args.pmi_batch):
                loss = uavtrack.pmi_contrastive_loss(pmi(x12), pmi(x13))
                opt_p.zero_grad(); loss.backward(); opt_p.step()
                lp = float(loss.detach())
            env.set_pmi(pmi.state_dict())                             # eval-mode (running-stat) BatchNorm is what gets folded
        rollout.sync_actor()                                          # new weights for the next rollout
        ep = res["ep_sums"]                                           # [B, 5]: sum_t mean_i reward, 3 terms, covered
        ret, cov = float(ep[:, 0].mean()), float(ep[:, 4].mean()) / args.steps
        history.append(ret)
        torch.cuda.synchronize()
        print(f"iter {it:3d}  episode return {ret:8.3f}  covered targets/step {cov:5.2f}  actor loss {la:+.4f}  "
              f"critic loss {lc:.4f}  pmi loss {lp:.4f}  rollout {t_roll * 1e3:6.1f} ms ({per_iter / t_roll / 1e9:.2f} G agent-steps/s)  "
              f"iteration {(time.perf_counter() - t0) * 1e3:6.1f} ms", flush=True)
    env.close()
    return history


if __name__ == "__main__":
    main()
