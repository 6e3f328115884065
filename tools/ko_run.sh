#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "pmi_against or pmi_reward" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -4 gpurun_out/pytest_gpu.log
