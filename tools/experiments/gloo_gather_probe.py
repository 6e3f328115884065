#!/usr/bin/env python3
"""Experiment (GPU box, two ranks sharing ONE GPU over gloo -- the rehearsal of bench.py --gpus 2 --backend gloo): where the
time of gather_transitions / gather_rollout_summary goes.  Start with
  UAVTRACK_BENCH_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/experiments/gloo_gather_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "marl-uavs-targets-tracking_amd")]
import torch, torch.distributed as dist
from uavtrack import sharding
dist.init_process_group("gloo")
r = dist.get_rank()
dev = torch.device("cuda:0")
K = 4096
blk = torch.randint(0, 1 << 30, (K, 28), dtype=torch.int32, device=dev)
ep = torch.rand(4096, 5, device=dev)
busy = torch.rand(4096, 4096, device=dev)


def t(fn, reps=5, load=False):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize(); dist.barrier()
        if load:                      # the other rank's GPU work queued while this rank exchanges: what a rollout loop looks like
            for _ in range(20):
                busy @ busy
        t0 = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t0) * 1e3)
    return sorted(out)[len(out) // 2]


host = blk.cpu()
gathered = torch.empty(2 * K, 28, dtype=torch.int32)
res = {
    "d2h 458 KB (.cpu())": t(lambda: blk.cpu()),
    "gloo all_gather of 458 KB host tensors": t(lambda: dist.all_gather_into_tensor(gathered, host)),
    "h2d 917 KB (.to(dev)) + sync": t(lambda: (gathered.to(dev), torch.cuda.synchronize())),
    "gather_rollout_summary (80 KB)": t(lambda: sharding.gather_rollout_summary(ep)),
    "d2h with 20 queued GEMMs of this rank in front": t(lambda: blk.cpu(), load=True),
    "gloo all_gather with queued GEMMs": t(lambda: dist.all_gather_into_tensor(gathered, host), load=True),
}
if r == 0:
    for k, v in res.items():
        print(f"{k:55s} {v:9.3f} ms", flush=True)
dist.barrier()
dist.destroy_process_group()
