#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5f.log
: > $OUT
A="--envs 8192 --n 50 --m 25 --dim 3 --T 200 --reset --warm 20 --reps 16"
for rep in 1 2; do
for lib in marl-uavs-targets-tracking_amd/uavtrack/libuavtrack.so build_variants/c4_uk6.so build_variants/c4_uu25.so; do
  UAVTRACK_LIB_OLDER_OK=1 timeout -k 10 200 python3 tools/sweep.py --lib $lib $A 2>&1 | grep "lib=" >> $OUT
done
done
timeout -k 10 400 python3 -m pytest tests -x -q -m gpu -k "closed or chunk or rollout or greedy or actor" > gpurun_out/r5f_tests.log 2>&1
echo "tests rc=$?" >> $OUT; tail -3 gpurun_out/r5f_tests.log >> $OUT
timeout -k 10 300 python3 - >> $OUT 2>&1 <<'PY'
import sys, time, torch
sys.path[:0] = [".", "marl-uavs-targets-tracking_amd"]
import uavtrack, bench
class A: pass
args = A(); args.n_uav, args.m_targets, args.dim, args.box, args.cooperative, args.reward, args.pmi_hidden = 20, 10, 2, 2000.0, 0.0, "raw", 128
args.policy, args.seed = "given", 42
torch.manual_seed(42)
actor = uavtrack.ActorMLP(hidden_dim=128, action_dim=12).to("cuda:0")
for mode in ("actor_chunks", "greedy_chunks"):
    env = bench.make_env(uavtrack, args, 4096, "cuda:0")
    ro = uavtrack.BatchedRollout(env, "greedy" if mode.startswith("greedy") else actor, steps_per_graph=10, use_graph=True,
                                 device_actor=mode.startswith("actor"), fuse_chunks=True)
    ro.reset(seed=42); ro.run(40); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ro.run(400); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(mode, "%.2f G, %.1f us per 10-step chunk" % (4096 * 20 * 400 / best / 1e9, best / 40 * 1e6))
PY
grep -v amdgpu.ids $OUT
