#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/dbg_cov.py 2>&1 | grep -v amdgpu | head -3
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
for B in 3072 4096; do timeout -k 10 120 python3 tools/sweep.py --envs $B --T 200 --reps 11 2>&1 | grep lib=; done
