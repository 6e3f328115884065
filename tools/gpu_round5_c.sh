#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5c.log
: > $OUT
timeout -k 10 200 python3 tools/experiments/stamp_run.py 4096 > gpurun_out/stamps.log 2>&1
echo "stamps rc=$?" >> $OUT
timeout -k 10 200 python3 tools/experiments/stamp_run.py 1024 >> gpurun_out/stamps.log 2>&1
timeout -k 10 200 python3 tools/experiments/stamp_run.py 768 >> gpurun_out/stamps.log 2>&1
grep -v amdgpu.ids gpurun_out/stamps.log >> $OUT
UAVTRACK_BENCH_ONE_GPU=1 timeout -k 10 200 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/experiments/gloo_gather_probe.py > gpurun_out/gloo_probe.log 2>&1
echo "gloo probe rc=$?" >> $OUT; grep " ms" gpurun_out/gloo_probe.log >> $OUT
grep -v amdgpu.ids $OUT
