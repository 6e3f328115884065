import sys, time, torch
sys.path[:0] = ["/root/repo", "/root/repo/marl-uavs-targets-tracking_amd"]
import uavtrack
cfg = uavtrack.EnvConfig(n_envs=4096, n_uav=20, m_targets=10)
actor = uavtrack.ActorMLP().cuda()
env = uavtrack.BatchedUavEnv(cfg)
ro = uavtrack.BatchedRollout(env, actor, steps_per_graph=10, use_graph=True)
ro.reset(seed=1); ro.run(40); torch.cuda.synchronize()
t0 = time.perf_counter(); ro.run(400); torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter()-t0)*1e3/400)
obs = ro.obs
def timeit(name, fn, n=50):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-t0)*1e6/n:.1f} us")
with torch.no_grad():
    timeit("actor", lambda: actor(obs))
    p = actor(obs)
    timeit("sample", lambda: uavtrack.sample_actions(p))
    a = uavtrack.sample_actions(p)
    timeit("env.step", lambda: env.step(a))
    o, r, d = env.step(a); info = env.info
    timeit("accum", lambda: (ro.ep[:,0].add_(r.mean(dim=1)), ro.ep[:,1:4].add_(info["terms"].mean(dim=2).t()), ro.ep[:,4].add_(info["covered"].float())))
