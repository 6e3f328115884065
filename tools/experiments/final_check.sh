#!/bin/bash
cd "${GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out
bash tools/gpu_check.sh > gpurun_out/gpu_check_stdout.log 2>&1
timeout -k 10 500 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench default rc=$?" >> gpurun_out/gpu_check.log
python3 tools/perf_floor.py gpurun_out/bench_default.json >> gpurun_out/gpu_check.log 2>&1
grep -v amdgpu.ids gpurun_out/gpu_check.log
