"""CPU-only checks of the host layer and of the C-ABI boundary: the library loads,
exports every symbol include/uavtrack.h declares, and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT

import uavtrack
from uavtrack import _lib
from uavtrack.config import EnvConfig, RewardMode
from uavtrack.pmi import fold_pmi_state_dict, pmi_blob_size
from uavtrack.sharding import shard_range


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "uavtrack.h")).read()
    declared = set(re.findall(r"\b(uavtrack_[a-z_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.uavtrack_version() == _lib.ABI_VERSION


def test_config_struct_matches_header_layout():
    # 12 int32/uint32 (48 B) + int64 + 14 doubles; no implicit padding surprises
    assert C.sizeof(_lib.UavtrackConfig) == 48 + 8 + 14 * 8
    c = EnvConfig(n_envs=3, n_uav=20, m_targets=10, cooperative=0.3).c_struct(0)
    assert c.struct_size == C.sizeof(_lib.UavtrackConfig)
    assert (c.n_envs, c.n_uav, c.m_targets, c.reward_mode) == (3, 20, 10, int(RewardMode.MEAN))
    assert c.norm_n_uav == 20 and c.norm_m_targets == 10


def test_create_rejects_bad_config_with_message():
    lib = _lib.load()
    bad = EnvConfig(n_envs=0).c_struct(0)
    h = C.c_void_p()
    assert lib.uavtrack_create(C.byref(bad), C.byref(h)) != 0
    assert b"n_envs" in lib.uavtrack_last_error()
    bad = EnvConfig(n_envs=1, dim=2, nc=3).c_struct(0)
    assert lib.uavtrack_create(C.byref(bad), C.byref(h)) != 0
    assert b"nc must be 1" in lib.uavtrack_last_error()
    bad = EnvConfig().c_struct(0)
    bad.struct_size = 8
    assert lib.uavtrack_create(C.byref(bad), C.byref(h)) != 0
    assert b"ABI mismatch" in lib.uavtrack_last_error()


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback_fails_loudly():
    lib = _lib.load()
    ok = EnvConfig(n_envs=4).c_struct(0)
    h = C.c_void_p()
    assert lib.uavtrack_create(C.byref(ok), C.byref(h)) != 0
    assert b"no CPU fallback" in lib.uavtrack_last_error()
    with pytest.raises(RuntimeError):
        uavtrack.BatchedUavEnv(EnvConfig(n_envs=4), "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback|no HIP device"):
        uavtrack.BatchedUavEnv(EnvConfig(n_envs=4), "cuda:0")


def test_null_handle_calls_return_errors():
    lib = _lib.load()
    assert lib.uavtrack_step(None, None, None, None, None, None, None, None) != 0
    assert b"null handle" in lib.uavtrack_last_error()
    assert lib.uavtrack_destroy(None) == 0


def test_reference_dict_adapter():
    ref_cfg = {"environment": {"n_uav": 10, "m_targets": 10, "x_max": 2000, "y_max": 2000, "na": 12},
               "uav": {"dt": 1, "v_max": 20, "h_max": 6, "dc": 500, "dp": 200, "alpha": 0.6, "beta": 0.2, "gamma": 0.2},
               "target": {"v_max": 5, "h_max": 6}, "cooperative": 0.3}
    c = EnvConfig.from_reference_dict(ref_cfg, n_envs=7)
    assert c == EnvConfig(n_envs=7, cooperative=0.3)
    assert abs(c.u_h_max - np.pi / 6) < 1e-15 and c.resolved_mode() == RewardMode.MEAN
    assert EnvConfig().resolved_mode() == RewardMode.RAW


def test_pmi_fold_equals_unfolded_eval_forward(pmi_state_dict):
    sd = pmi_state_dict
    blob, H = fold_pmi_state_dict(sd)
    assert H == 128 and blob.size == pmi_blob_size(128) and blob.dtype == np.float32
    rng = np.random.RandomState(0)
    x = rng.randn(16, 12)

    def bn(v, p):
        return (v - sd[p + ".running_mean"]) / np.sqrt(sd[p + ".running_var"] + 1e-5) * sd[p + ".weight"] + sd[p + ".bias"]
    cat = np.concatenate([
        np.maximum(bn(x[:, :5] @ sd["fc_comm.weight"].T + sd["fc_comm.bias"], "bn_comm"), 0),
        np.maximum(bn(x[:, 5:9] @ sd["fc_obs.weight"].T + sd["fc_obs.bias"], "bn_obs"), 0),
        np.maximum(bn(x[:, 9:] @ sd["fc_boundary_state.weight"].T + sd["fc_boundary_state.bias"], "bn_boundary_state"), 0)], 1)
    hid = np.maximum(bn(cat @ sd["fc1.weight"].T + sd["fc1.bias"], "bn1"), 0)
    want = hid @ sd["fc2.weight"].T + sd["fc2.bias"]

    b = blob.astype(np.float64)
    o = 0
    def take(n, shape):
        nonlocal o
        v = b[o:o + n].reshape(shape); o += n
        return v
    wc, bc = take(5 * H, (5, H)), take(H, (H,))
    wo, bo = take(4 * H, (4, H)), take(H, (H,))
    wb, bb = take(3 * H, (3, H)), take(H, (H,))
    w1, b1 = take(3 * H * H, (3 * H, H)), take(H, (H,))
    w2, b2 = take(H, (H,)), take(1, (1,))
    cat2 = np.concatenate([np.maximum(x[:, :5] @ wc + bc, 0), np.maximum(x[:, 5:9] @ wo + bo, 0),
                           np.maximum(x[:, 9:] @ wb + bb, 0)], 1)
    got = np.maximum(cat2 @ w1 + b1, 0) @ w2 + b2
    np.testing.assert_allclose(got, want[:, 0], rtol=0, atol=2e-5)


def test_shard_range_partitions_exactly():
    for total in (1, 7, 8, 4096, 32768, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
    with pytest.raises(ValueError):
        shard_range(8, 8, 8)


def test_export_matches_reference_csv_layout(tmp_path):
    """environment.py:229-244 layouts: u_xy rows are UAV-major then step; one covered count per step."""
    from uavtrack.export import save_covered_num, save_uav_positions, uav_tracks_from_obs
    T, B, N = 4, 3, 2
    obs = np.zeros((T, B, N, 12), dtype=np.float32)
    for t in range(T):
        for i in range(N):
            obs[t, 1, i, 9] = (100 * i + t) / 500.0       # x / dc
            obs[t, 1, i, 10] = (7 * i + 2 * t) / 500.0    # y / dc
    xs, ys = uav_tracks_from_obs(obs, 500.0, env_index=1)
    assert xs.shape == (T, N) and abs(xs[3, 1] - 103) < 1e-3 and abs(ys[2, 1] - 11) < 1e-3
    path = save_uav_positions(str(tmp_path), 5, xs, ys)
    rows = np.loadtxt(path, delimiter=",", skiprows=1)
    # what the reference computes: np.array([all_uav_xs, all_uav_ys]).transpose().reshape(-1, 2)
    want = np.array([xs.tolist(), ys.tolist()]).transpose().reshape(-1, 2)
    np.testing.assert_allclose(rows, want, atol=1e-3)
    assert open(path).readline().strip() == "x,y" and rows.shape == (N * T, 2)
    np.testing.assert_allclose(rows[:T, 0], [0, 1, 2, 3], atol=1e-3)        # UAV 0 over the steps first
    cov = np.arange(T * B).reshape(T, B)
    cpath = save_covered_num(str(tmp_path), 5, cov, env_index=2)
    np.testing.assert_array_equal(np.loadtxt(cpath, skiprows=1), cov[:, 2])


def test_replay_buffers_reference_semantics():
    """DeviceReplayBuffer / PrioritizedDeviceReplayBuffer against the semantics of train.py:41-139 (deque(maxlen)
    FIFO, sampling without / with replacement, max-priority insertion, importance weights), on CPU tensors."""
    import collections
    import torch
    from uavtrack import DeviceReplayBuffer, PrioritizedDeviceReplayBuffer, transitions_from_rollout
    g = torch.Generator().manual_seed(0)
    T, B, N = 5, 3, 4
    obs_in = torch.randn(B, N, 12, generator=g)
    out = {"obs": torch.randn(T, B, N, 12, generator=g), "actions": torch.randint(0, 12, (T, B, N), generator=g).int(),
           "reward": torch.randn(T, B, N, generator=g)}
    tr = transitions_from_rollout(obs_in, out)
    n = T * B * N
    assert tr["states"].shape == (n, 12) and torch.equal(tr["states"][:B * N], obs_in.reshape(-1, 12))
    assert torch.equal(tr["states"][B * N:], out["obs"][:-1].reshape(-1, 12)) and torch.equal(tr["next_states"], out["obs"].reshape(-1, 12))
    # FIFO overwrite == deque(maxlen)
    cap = 37
    buf = DeviceReplayBuffer(cap, "cpu")
    ref = collections.deque(maxlen=cap)
    for rep in range(3):
        buf.add(tr)
        ref.extend(zip(tr["states"], tr["actions"], tr["rewards"], tr["next_states"]))
        assert buf.size() == len(ref)
    order = [(buf.pos + k) % cap for k in range(cap)]            # oldest -> newest
    for slot, (s, a, r, s2) in zip(order, ref):
        assert torch.equal(buf.store["states"][slot], s) and buf.store["actions"][slot] == a
        assert buf.store["rewards"][slot] == r and torch.equal(buf.store["next_states"][slot], s2)
    smp = buf.sample(20, generator=g)
    assert smp["states"].shape == (20, 12) and smp["actions"].dtype == torch.int32
    rows = {tuple(x.tolist()) for x in buf.store["states"]}
    assert all(tuple(x.tolist()) in rows for x in smp["states"])
    assert len({tuple(x.tolist()) for x in smp["states"]}) == 20      # without replacement
    assert buf.sample(1000)["actions"].numel() == cap                 # min(batch, size)
    # one add larger than the capacity keeps the newest `capacity` transitions
    big = DeviceReplayBuffer(10, "cpu")
    big.add(tr)
    assert big.size() == 10
    newest = {tuple(x.tolist()) for x in tr["states"][-10:]}
    assert {tuple(x.tolist()) for x in big.store["states"]} == newest
    # prioritised
    pb = PrioritizedDeviceReplayBuffer(64, "cpu", alpha=0.6)
    assert pb.sample(4)[1] is None
    pb.add({k: v[:16] for k, v in tr.items()})
    assert torch.all(pb.priorities[:16] == 1.0) and torch.all(pb.priorities[16:] == 0.0)
    pb.update_priorities(torch.tensor([3, 5]), torch.tensor([9.0, 0.25]))
    pb.add({k: v[16:20] for k, v in tr.items()})
    assert torch.all(pb.priorities[16:20] == 9.0)                     # new entries take the current maximum
    smp, idx, w = pb.sample(4000, beta=0.4, generator=g)
    assert idx.numel() == 20 and w.max() == 1.0                      # min(batch, size), with replacement
    smp, idx, w = PrioritizedDeviceReplayBuffer.sample(pb, 20, 0.4, g)
    prob = pb.priorities[:20] ** 0.6
    prob = prob / prob.sum()
    want_w = (20 * prob[idx]) ** -0.4
    np.testing.assert_allclose(w.numpy(), (want_w / want_w.max()).numpy(), rtol=1e-6)
    many = torch.multinomial(prob, 20000, replacement=True, generator=g)
    freq = torch.bincount(many, minlength=20).double() / 20000
    assert abs(freq[3] - prob[3].double()) < 0.01 and freq[3] > 3 * freq[0]


def test_pmi_training_data_path():
    """sample_pmi_pairs == the per-row copy loop of PMINet.py:78-84 for the same indices; loss == CustomLoss."""
    import torch
    from uavtrack import sample_pmi_pairs, pmi_contrastive_loss, pmi_batches
    g = torch.Generator().manual_seed(1)
    T, N = 11, 6
    data = torch.randn(T * N, 12, generator=g)
    sel, t_idx, u_idx = sample_pmi_pairs(data, N, 50, generator=g)
    assert sel.shape == (50, 2, 12) and t_idx.max() < T and u_idx.max() < N
    view = data.view(T, N, 12)
    want = torch.zeros(50, 2, 12)
    for i in range(50):                                   # the reference's loop
        want[i] = view[t_idx[i], u_idx[i]]
    assert torch.equal(sel, want)
    o1, o2 = torch.randn(32, 1, generator=g) * 3, torch.randn(32, 1, generator=g) * 3
    ref_loss = torch.mean(torch.log(1 + torch.exp(-o1)) + torch.log(1 + torch.exp(o2)))
    assert torch.allclose(pmi_contrastive_loss(o1, o2), ref_loss, rtol=1e-6)
    assert torch.isfinite(pmi_contrastive_loss(torch.tensor([-200.0]), torch.tensor([200.0])))   # the naive form overflows
    batches = list(pmi_batches(sel, 16))
    assert len(batches) == 3 and batches[0][0].shape == (16, 12) and torch.equal(batches[1][1], sel[16:32, 1])


def test_pmi_training_path_pinned_to_reference_train_pmi():
    """f3 against the reference itself: tests/golden/f3_pmi_train.npz holds what PMINetwork.train_pmi
    (PMINet.py:74-100) really fed to forward() -- its torch.randint triples, per-row copy and batch slicing -- under
    torch.manual_seed(123) on a recorded observation history, the outputs it got and the avg_loss it returned
    (oracle/gen_golden.py: gen_pmi_train).  Under the same seed sample_pmi_pairs + pmi_batches must produce those very
    batches, CustomLoss those losses, and train_pmi_epoch on the same initial weights the same training trajectory."""
    import torch
    from conftest import load_golden
    from uavtrack import sample_pmi_pairs, pmi_contrastive_loss, pmi_batches, train_pmi_epoch, make_pmi_net
    z, meta = load_golden("f3_pmi_train")
    N, b2, bs = meta["n_uav"], meta["b2_size"], meta["batch_size"]
    data = torch.from_numpy(z["train_data"])
    assert data.shape == (meta["steps"] * N, 12)
    torch.manual_seed(meta["torch_seed"])
    sel, t_idx, u_idx = sample_pmi_pairs(data, N, b2)
    batches = list(pmi_batches(sel, bs))
    assert len(batches) == b2 // bs == z["in_1_2"].shape[0]
    for k, (x12, x13) in enumerate(batches):
        np.testing.assert_array_equal(x12.numpy(), z["in_1_2"][k])
        np.testing.assert_array_equal(x13.numpy(), z["in_1_3"][k])
    # the [T, N, 12] form of the same history (what a device rollout returns) selects the same rows
    torch.manual_seed(meta["torch_seed"])
    sel3, _, _ = sample_pmi_pairs(data.view(meta["steps"], N, 12), N, b2)
    assert torch.equal(sel3, sel)
    # CustomLoss on the reference's recorded outputs -> the avg_loss it returned (mean of |loss| over the batches)
    losses = [float(pmi_contrastive_loss(torch.from_numpy(z["out_1_2"][k]), torch.from_numpy(z["out_1_3"][k])))
              for k in range(len(batches))]
    assert abs(np.mean(np.abs(losses)) - float(z["avg_loss"])) < 1e-6
    # the whole call: same initial weights (PMINetwork's constructor order and init under manual_seed(11)), Adam(1e-3)
    torch.manual_seed(11)
    net = make_pmi_net(meta["hidden"])
    opt = torch.optim.Adam(net.parameters(), lr=0.001)        # PMINet.py:39
    seen = []
    hook = net.register_forward_hook(lambda m, i, o: seen.append((i[0].detach().clone(), o.detach().clone())))
    torch.manual_seed(meta["torch_seed"])
    avg = train_pmi_epoch(net, opt, data, N, b2, bs)
    hook.remove()
    assert abs(avg - float(z["avg_loss"])) < 1e-5
    assert len(seen) == 2 * len(batches)
    for k in range(len(batches)):
        np.testing.assert_array_equal(seen[2 * k][0].numpy(), z["in_1_2"][k])
        np.testing.assert_allclose(seen[2 * k][1].numpy(), z["out_1_2"][k], rtol=0, atol=2e-5)     # weights after k Adam steps
        np.testing.assert_allclose(seen[2 * k + 1][1].numpy(), z["out_1_3"][k], rtol=0, atol=2e-5)


def _ref_cfg(n, m, x_max=2000, y_max=2000, na=12):
    return {"environment": {"n_uav": n, "m_targets": m, "x_max": x_max, "y_max": y_max, "na": na},
            "uav": {"dt": 1, "v_max": 20, "h_max": 6, "dc": 500, "dp": 200, "alpha": 0.6, "beta": 0.2, "gamma": 0.2},
            "target": {"v_max": 5, "h_max": 6}, "cooperative": 0}


def test_reference_reset_and_step_draws_are_seed_identical():
    """north_star "on identical seeds": the B = 1 adapter draws its reset from Python's global `random` in the reference's
    order (environment.py:54-83) and burns the reference's M unused draws per step (target.py:34).  Host half of that,
    against what the reference itself recorded: g6's four reset layouts and g1's initial state EXACTLY (fp64), and g1's 200 x 5
    `random.randint` actions exactly -- they were drawn between the steps of the reference run, so they only come out
    right if every step has consumed what the reference's step consumes."""
    import random
    from conftest import load_golden
    from uavtrack.compat import reference_reset_draw, reference_step_draws
    z6, meta6 = load_golden("g6_reset")
    for tag, mm in meta6.items():
        n, m = mm["n_uav"], mm["m_targets"]
        random.seed(42)
        st = reference_reset_draw(_ref_cfg(n, m), n, m, 2000, 2000, 12)
        for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th"):
            np.testing.assert_array_equal(st[k], z6[f"{tag}_{k}"], err_msg=f"{tag} {k}")
    z1, meta1 = load_golden("g1_n5m3_raw")
    cfg = _ref_cfg(5, 3)
    random.seed(meta1["seeds"][0])
    st = reference_reset_draw(cfg, 5, 3, 2000, 2000, 12)
    for k in ("ux", "uy", "uh", "ua", "tx", "ty", "th"):
        np.testing.assert_array_equal(st[k], z1[k][0, 0], err_msg=k)
    for t in range(meta1["steps"]):
        a = [random.randint(0, 11) for _ in range(5)]
        assert a == [int(v) for v in z1["actions"][0, t]], t
        reference_step_draws(cfg, 3)
    # a config with FEWER UAVs than the environment: the reference raises IndexError (environment.py:59) -- refused here too
    with pytest.raises(ValueError):
        reference_reset_draw(_ref_cfg(4, 3), 5, 3, 2000, 2000, 12)
    # ... with MORE: the reference lays out n_cfg start positions, spaced by x_max / (n_cfg + 1), and uses the first n_uav
    # (environment.py:54-59, 105).  Expected values: the unmodified reference, Environment(n_uav=5) reset with a config
    # that says n_uav = 6, after random.seed(42)
    random.seed(42)
    st = reference_reset_draw(_ref_cfg(6, 3), 5, 3, 2000, 2000, 12)
    np.testing.assert_array_equal(st["ux"], [i * 2000 / 7 for i in range(1, 6)])
    np.testing.assert_allclose(st["uh"][:2], [0.8760444114976647, 1.5177065510328678], rtol=0, atol=1e-15)
    np.testing.assert_allclose(st["tx"], [1180.9850248980792, 1204.0374580999608, 839.0396419233175], rtol=0, atol=1e-12)


def test_greedy_seed_is_pinned_by_random_seed_across_processes():
    """The C-METHOD path of the adapter keys the library's Philox draws with a seed derived from the global generator's
    state (compat.seed_from_global_random).  It must be the same in every process after the same random.seed -- a hash of
    the state tuple is not (it ends in None, whose hash is an address on CPython 3.10) -- and must not consume a draw."""
    import random
    import subprocess
    import sys
    from uavtrack.compat import seed_from_global_random
    code = ("import sys; sys.path[:0] = %r; import random; from uavtrack.compat import seed_from_global_random; "
            "random.seed(42); print(seed_from_global_random())" % ([ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")],))
    outs = [subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.strip() for _ in range(2)]
    random.seed(42)
    here = seed_from_global_random()
    assert outs[0] == outs[1] == str(here), (outs, here)
    assert 0 <= here < 2 ** 63
    nxt = random.random()
    random.seed(42)
    assert random.random() == nxt          # the derivation left the generator where random.seed put it
    random.seed(43)
    assert seed_from_global_random() != here


def build_abi_client(tmp_path):
    """gcc (C99, no C++, no Python) against include/uavtrack.h + libuavtrack.so + the HIP runtime."""
    import subprocess
    libdir = os.path.join(ROOT, "marl-uavs-targets-tracking_amd", "uavtrack")
    exe = os.path.join(str(tmp_path), "abi_roundtrip")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "abi", "abi_roundtrip.c"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64",
                    "-L" + libdir, "-luavtrack", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + libdir], check=True)
    return exe


def test_c_abi_builds_from_plain_c_and_fails_loudly_without_gpu(tmp_path):
    """The boundary is a real C ABI: a C99 translation unit compiles against the header with -Werror, links,
    and -- in this GPU-less container -- gets status != 0 plus uavtrack_last_error(), not a fallback."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test")
    exe = build_abi_client(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 3, (r.returncode, r.stderr)
    assert "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_launch_plan_and_cores():
    """bench.launch_plan: at most `rollout` steps per launch, never across an episode end, sums to `steps`;
    describe_plan names exactly those launches (the driver's --steps 20 --warmup 5 is ONE 20-step launch)."""
    bench = _load_bench()
    for steps, rollout, horizon, pos in ((2000, 200, 200, 0), (1000, 1, 200, 0), (950, 64, 200, 130), (7, 200, 200, 199)):
        plan, end = bench.launch_plan(steps, rollout, horizon, pos)
        assert sum(plan) == steps and max(plan) <= rollout and min(plan) >= 1
        p = pos
        for T in plan:
            assert p + T <= horizon          # a launch never spans two episodes
            p = (p + T) % horizon
        assert end == p
    warm, pos = bench.launch_plan(5, 200, 200, 0)
    timed, _ = bench.launch_plan(20, 200, 200, pos)
    assert warm == [5] and timed == [20] and bench.describe_plan(timed) == "1 x 20 steps"
    assert bench.describe_plan([195, 200, 200, 5]) == "1 x 195 + 2 x 200 + 1 x 5 steps"
    assert 1 <= bench.host_cores() <= os.cpu_count()
    # the end-of-rollout gathers of a plan: behind every launch that ends an episode, and behind the last launch in any case
    assert bench.gather_points([20], 5, 200) == [0]
    assert bench.gather_points([200] * 10, 0, 200) == list(range(10))
    assert bench.gather_points([195, 200, 200, 5], 5, 200) == [0, 1, 2, 3]
    assert bench.gather_points([64, 64, 64, 8, 64], 0, 200) == [3, 4]
    assert bench.gather_points([1] * 400, 100, 200) == [99, 299, 399]
    assert bench.gather_points([], 0, 200) == []


def test_bench_gpus_n_starts_n_ranks_or_fails():
    """`python bench.py --gpus N` (no torchrun) starts N ranks itself: the launcher self-test (no GPU: gloo
    rendezvous + all-reduce) reports n_gpus == rccl_world_size == 2 on ONE line; without enough GPUs, or with a
    WORLD_SIZE that disagrees with --gpus, the run exits non-zero instead of reporting fewer GPUs."""
    import json
    import subprocess
    import sys
    exe = [sys.executable, os.path.join(ROOT, "bench.py")]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(exe + ["--gpus", "2", "--backend", "gloo", "--selftest-launcher"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world_size"] == 2 and d["gpus_requested"] == 2
    # with the driver's flags the timed region of an N > 1 run is ONE 20-step launch and exactly ONE
    # end-of-rollout gather behind it (the rollout IS the K steps); the default flags gather behind every episode
    r = subprocess.run(exe + ["--gpus", "2", "--backend", "gloo", "--selftest-launcher", "--steps", "20", "--warmup", "5"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["timed_launch_steps"] == [20] and d["gather"] == {"in_region": 1, "behind_launches": [0]}
    # a rank count that cannot be honoured is an error, never a silent 1-GPU run
    r = subprocess.run(exe + ["--gpus", "2", "--selftest-launcher"], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()
    if not _have_gpu():
        r = subprocess.run(exe + ["--gpus", "2", "--steps", "5", "--warmup", "1"], capture_output=True, text=True,
                           timeout=120, env=env)
        assert r.returncode != 0 and "refusing" in r.stderr and not r.stdout.strip()


def test_pmi_net_is_state_dict_compatible_and_folds(pmi_state_dict):
    """make_pmi_net: the reference PMINetwork's recorded state_dict (golden pmi_h128) loads strictly; its eval-mode
    forward equals the BatchNorm-folded affine chain that set_pmi uploads (fold_pmi_state_dict)."""
    import torch
    from uavtrack import make_pmi_net
    net = make_pmi_net(128)
    sd = {k: (torch.as_tensor(np.asarray(v))) for k, v in pmi_state_dict.items()}
    for bn in ("bn_comm", "bn_obs", "bn_boundary_state", "bn1"):
        sd.setdefault(bn + ".num_batches_tracked", torch.tensor(0))
    net.load_state_dict(sd, strict=True)
    net.eval()
    x = torch.randn(64, 12, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = net(x).double().numpy().ravel()
    blob, H = fold_pmi_state_dict(net.state_dict())
    assert H == 128
    b = blob.astype(np.float64)
    o = 0

    def take(n):
        nonlocal o
        v = b[o:o + n]; o += n
        return v
    wc, bc = take(5 * H).reshape(5, H), take(H)
    wo, bo = take(4 * H).reshape(4, H), take(H)
    wb, bb = take(3 * H).reshape(3, H), take(H)
    w1, b1 = take(3 * H * H).reshape(3 * H, H), take(H)
    w2, b2 = take(H), take(1)
    xn = x.double().numpy()
    a = np.concatenate([np.maximum(xn[:, 0:5] @ wc + bc, 0), np.maximum(xn[:, 5:9] @ wo + bo, 0),
                        np.maximum(xn[:, 9:12] @ wb + bb, 0)], axis=1)
    got = np.maximum(a @ w1 + b1, 0) @ w2 + b2
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)


def test_x6_split_error_model_on_adversarial_sums():
    """The arithmetic of pmi_score_x6_kernel (csrc/pmi_kernel.hip) emulated in numpy (tools/x6_accuracy.py, now part of
    the suite): fp32 = three bf16 parts by truncation; a product is the six bf16 products of total order <= 2, exact, with
    fp32 accumulation per 16-wide k-step.  On well-scaled data AND on the adversarial layer of conftest (magnitudes over
    2^-20 .. 2^4, every 384-term sum a ~1000-fold cancellation) the six-term scheme is as accurate as an fp32 fmaf chain
    (what the fp32 MFMA and the reference's fp32 torch deliver) and within 1e-5 of sum |terms|; three terms are not
    enough.  The kernel itself is checked against fp64 on the GPU (tests/test_hip_round3.py)."""
    from conftest import adversarial_pmi_state_dict

    def trunc(a):
        return (a.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)

    def split3(a):
        h = trunc(a)
        r1 = (a - h).astype(np.float32)
        m = trunc(r1)
        return h, m, trunc((r1 - m).astype(np.float32))

    def mm(a, b, acc=None):
        acc = np.zeros((a.shape[0], b.shape[1]), np.float32) if acc is None else acc
        for k in range(0, a.shape[1], 16):           # one MFMA: exact products, one fp32 rounding of the k-step's sum
            acc = (acc.astype(np.float64) + a[:, k:k + 16].astype(np.float64) @ b[k:k + 16].astype(np.float64)).astype(np.float32)
        return acc

    rng = np.random.RandomState(0)
    K, N, M = 384, 128, 256
    cases = {"well_scaled": (np.maximum(rng.randn(M, K).astype(np.float32) * 3, 0), (rng.randn(K, N) * 0.1).astype(np.float32))}
    sd = adversarial_pmi_state_dict(128, seed=1)
    h = np.repeat(np.abs(rng.randn(M, K // 2)).astype(np.float32) * 2, 2, axis=1)      # twin activations
    cases["adversarial"] = (h, np.ascontiguousarray(sd["fc1.weight"].T))
    for name, (x, w) in cases.items():
        ref = x.astype(np.float64) @ w.astype(np.float64)
        mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)
        xh, xm, xl = split3(x)
        wh, wm, wl = split3(w)
        # the kernel's order: small terms first, ONE accumulation chain (pmi_score_x6_kernel, t = 0..5 per k-step)
        six = np.zeros((M, N), np.float32)
        three = np.zeros((M, N), np.float32)
        for k in range(0, K, 16):
            sl = slice(k, k + 16)
            for a, b in ((xh, wl), (xl, wh), (xm, wm), (xm, wh), (xh, wm), (xh, wh)):
                six = mm(a[:, sl], b[sl], six)
            for a, b in ((xm, wh), (xh, wm), (xh, wh)):
                three = mm(a[:, sl], b[sl], three)
        chain = np.zeros((M, N), np.float32)
        for k in range(K):
            chain = (chain.astype(np.float64) + x[:, k:k + 1].astype(np.float64) * w[k:k + 1].astype(np.float64)).astype(np.float32)
        e6, e3, ec = np.abs(six - ref), np.abs(three - ref), np.abs(chain - ref)
        assert (e6 <= 1e-5 * np.maximum(mag, 1.0)).all(), (name, e6.max(), mag.max())
        assert e6.max() <= 2.0 * ec.max() + 1e-6 * mag.max(), (name, e6.max(), ec.max())
        assert e3.max() > 8.0 * e6.max(), (name, e3.max(), e6.max())       # why six products, not three


def test_t3_block_scaled_f16_split_error_model():
    """The arithmetic of pmi_score_t3_kernel's 3H x H layer emulated in numpy: operands block-scaled by powers of two (S on
    the activations from a BOUND that may be far above the values, T on the weights from max |w|), x = f16(x) (toward
    zero) + f16(x - hi) with NO scaling of the remainder -- it is a normal f16 number for every value within 2^-12 of the
    bound, subnormal (a few bits short) below -- three exact products per fp32 product, two fp32 accumulators, small terms
    first.  Against fp64 it is as accurate as an fp32 fmaf chain on well-scaled data, on the adversarial cancellation
    layer of conftest, on activations 1000 times smaller than typical, and with the bound 2^12 above the largest value."""
    from conftest import adversarial_pmi_state_dict

    def rtz16(a):
        h = a.astype(np.float16)
        over = np.abs(h.astype(np.float32)) > np.abs(a)
        return np.where(over, np.nextafter(h, np.float16(0)), h)

    def mm(a, b, acc):
        for k in range(0, a.shape[1], 16):           # one MFMA: exact products, one fp32 rounding of the k-step's sum
            acc = (acc.astype(np.float64) + a[:, k:k + 16].astype(np.float64) @ b[k:k + 16].astype(np.float64)).astype(np.float32)
        return acc

    rng = np.random.RandomState(0)
    K, N, M = 384, 128, 128
    cases = {"well_scaled": (np.maximum(rng.randn(M, K).astype(np.float32) * 3, 0), (rng.randn(K, N) * 0.1).astype(np.float32)),
             "tiny_activations": (np.maximum(rng.randn(M, K).astype(np.float32) * 3e-3, 0), (rng.randn(K, N) * 0.1).astype(np.float32))}
    sd = adversarial_pmi_state_dict(128, seed=1)
    h = np.repeat(np.abs(rng.randn(M, K // 2)).astype(np.float32) * 2, 2, axis=1)
    cases["adversarial"] = (h, np.ascontiguousarray(sd["fc1.weight"].T))
    for name, (x, w) in cases.items():
        ref = x.astype(np.float64) @ w.astype(np.float64)
        mag = np.abs(x).astype(np.float64) @ np.abs(w).astype(np.float64)
        chain = np.zeros((M, N), np.float32)
        for k in range(K):
            chain = (chain.astype(np.float64) + x[:, k:k + 1].astype(np.float64) * w[k:k + 1].astype(np.float64)).astype(np.float32)
        ec = np.abs(chain - ref)
        for slack in (1.0, 64.0, 4096.0):             # the activation bound over the largest activation
            S = np.float32(2.0 ** np.floor(np.log2(512.0 / (np.abs(x).max() * slack))))
            T = np.float32(2.0 ** np.floor(np.log2(32000.0 / np.abs(w).max())))
            xs, ws = x * S, w * T                      # (powers of two: exact)
            xh = rtz16(xs)
            xl = (xs - xh.astype(np.float32)).astype(np.float16)
            wh = ws.astype(np.float16)
            wl = (ws - wh.astype(np.float32)).astype(np.float16)
            f = lambda a: a.astype(np.float32)
            acc_l, acc_h = np.zeros((M, N), np.float32), np.zeros((M, N), np.float32)
            for k in range(0, K, 16):
                sl = slice(k, k + 16)
                acc_l = mm(f(xh[:, sl]), f(wl[sl]), acc_l)
                acc_l = mm(f(xl[:, sl]), f(wh[sl]), acc_l)
                acc_h = mm(f(xh[:, sl]), f(wh[sl]), acc_h)
            got = (acc_h + acc_l) * np.float32(1.0 / (S * T))
            e = np.abs(got - ref)
            assert (e <= 1e-5 * np.maximum(mag, 1e-30)).all(), (name, slack, (e / np.maximum(mag, 1e-30)).max())
            assert e.max() <= 2.0 * ec.max() + 1e-6 * mag.max(), (name, slack, e.max(), ec.max())


def test_clip_overstep_diagnostics():
    """The reference prints "overstep in clip." when a raw term leaves its range (data_util.py:44-47); the kernels clamp.
    `EnvConfig.clip_can_overstep` says from the constants alone which of the four clips can trigger (the reference's own
    configurations: none), `BatchedUavEnv.clip_saturation` counts the outputs that sit on a bound -- checked here on the
    oracle's outputs for a configuration whose tracking ceiling is below what ten targets can add up to."""
    import torch
    import uavtrack
    from oracle import OracleConfig, OracleEnv
    assert uavtrack.EnvConfig(n_uav=20, m_targets=10).clip_can_overstep() == {"tracking": False, "duplicate": False, "boundary": False, "reward": False}
    c = uavtrack.EnvConfig(n_uav=20, m_targets=10, norm_m_targets=1, norm_n_uav=5, alpha=1.5)
    assert c.clip_can_overstep() == {"tracking": True, "duplicate": True, "boundary": False, "reward": True}
    assert uavtrack.EnvConfig(n_uav=6, m_targets=3, norm_n_uav=5).clip_can_overstep()["duplicate"] is False      # 5 terms against 5
    B, N, M = 16, 20, 10
    orc = OracleEnv(OracleConfig(n_envs=B, n_uav=N, m_targets=M, norm_m_targets=1, x_max=300.0, y_max=300.0), n_threads=2)
    orc.reset_philox(3)
    rng = np.random.RandomState(0)
    terms, rewards = [], []
    for _ in range(5):
        out = orc.step(rng.randint(0, 12, size=(B, N)).astype(np.int32))
        terms.append(out["terms"]); rewards.append(out["reward"])
    terms, rewards = np.stack(terms), np.stack(rewards)
    got = uavtrack.BatchedUavEnv.clip_saturation(torch.from_numpy(terms), torch.from_numpy(rewards))
    assert got["tracking"] == int((terms[:, 0] >= 1.0).sum()) > 0 and got["duplicate"] == 0 and got["reward"] == 0


def test_profile_summary_files_the_scorer_that_moved_the_bytes(tmp_path):
    """a MAAC-R chunk launches TWO pmi_score* kernels -- the scorer that does the work and the gated stand-by,
    which returns at once.  tools/summarise_profile.py must file the HBM traffic of the first under the `_scorer` key of
    profiles/traffic.json (round 4 filed the stand-by's 36 864 B).  Synthetic rocprofv3 counter CSVs, run in a scratch cwd."""
    import csv
    import json
    import shutil
    import subprocess
    import sys
    work = tmp_path / "w"
    (work / "profiles").mkdir(parents=True)
    (work / "tools").mkdir()
    shutil.copy(os.path.join(ROOT, "tools", "summarise_profile.py"), work / "tools" / "summarise_profile.py")
    t3 = "void uavtrack::(anonymous namespace)::pmi_score_t3_kernel<128>(uavtrack::(anonymous namespace)::PmiParams)"
    x6 = "void uavtrack::(anonymous namespace)::pmi_score_x6_kernel<128>(uavtrack::(anonymous namespace)::PmiParams)"
    ro = "void uavtrack::(anonymous namespace)::rollout_kernel<20, 10, 2, false, 0, true, false, true>(uavtrack::StepParams)"
    cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "VGPR_Count", "SGPR_Count", "LDS_Block_Size",
            "Scratch_Size", "Workgroup_Size", "Grid_Size"]
    for counter, vals in (("FETCH_SIZE", {ro: 34000.0, t3: 113000.0, x6: 18.0}), ("WRITE_SIZE", {ro: 1261000.0, t3: 9490.0, x6: 0.0})):
        d = work / "prof" / f"pmc_{counter.lower()}" / "run"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols)
            for k, (name, v) in enumerate(vals.items()):       # the stand-by comes LAST, as in a real trace
                w.writerow([k + 1, name, counter, v, 128, 96, 0, 0, 256, 65536])
    r = subprocess.run([sys.executable, "tools/summarise_profile.py", "rXX", "prof", "4096x20x10_T200_pmi128"], cwd=work,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    tr = json.load(open(work / "profiles" / "traffic.json"))
    sc = tr["4096x20x10_T200_pmi128_scorer"]
    assert sc["kernel"] == "pmi_score_t3_kernel<128>" and abs(sc["hbm_bytes_per_launch"] - (2 * 113000.0 + 9490.0) * 1024) < 1.0
    assert tr["4096x20x10_T200_pmi128"]["kernel"].startswith("rollout_kernel<20, 10, 2")


def test_host_step_struct_mirrors_the_header():
    """struct uavtrack_host_step (include/uavtrack.h) and its ctypes mirror list the same pointer members in the same order:
    the adapter reads the library's host block through them."""
    hdr = open(os.path.join(ROOT, "include", "uavtrack.h")).read()
    body = re.search(r"typedef struct uavtrack_host_step \{(.*?)\} uavtrack_host_step;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        for m in re.finditer(r"\*\s*(\w+)", decl):
            names.append(m.group(1))
    assert names == [k for k, _ in _lib.HostStep._fields_], names
    assert C.sizeof(_lib.HostStep) == 8 * len(names)
    # and the entry point's signature: handle, host actions, the struct, the stream
    res, args = _lib.SIGNATURES["uavtrack_step_host"]
    assert res is C.c_int and len(args) == 4 and args[2] == C.POINTER(_lib.HostStep)
