#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/final_r02.log
: > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT
tail -4 gpurun_out/pytest_gpu.log >> $OUT
timeout -k 10 600 python3 tests/fuzz_api.py 60 > gpurun_out/fuzz.log 2>&1
echo "fuzz rc=$? $(tail -1 gpurun_out/fuzz.log)" >> $OUT
cat $OUT
PROFILE_MFMA=1 bash tools/profile_gpu.sh r02pmi --reward pmi --steps 1000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/prof_r02pmi.log 2>&1
echo "profile rc=$?"
