#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/chk_r02.log
: > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT
tail -3 gpurun_out/pytest_gpu.log >> $OUT
timeout -k 10 120 python3 tools/sweep.py --envs 4096 --T 200 --reps 9 >> $OUT 2>&1 &&
timeout -k 10 120 python3 tools/sweep.py --envs 65536 --T 50 --reps 7 >> $OUT 2>&1 &&
timeout -k 10 120 python3 tools/sweep.py --envs 8192 --n 50 --m 25 --dim 3 --T 50 --reps 7 >> $OUT 2>&1
cat $OUT
