#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 tests/fuzz_api.py 400 > gpurun_out/fuzz_api.log 2>&1; echo "fuzz_api rc=$?"; tail -4 gpurun_out/fuzz_api.log
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
