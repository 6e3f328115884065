#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/chk_r02.log
: > $OUT
UAVTRACK_TEST_REPORT=1 timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -s > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT; tail -5 gpurun_out/pytest_gpu.log >> $OUT
grep "knife-edge" gpurun_out/pytest_gpu.log | awk '{print $2, $3, $4, $5, $6, $7, $8}' | sort -t= -k2 -n -r | head -40 >> $OUT
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 3 --no-extras > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; echo "bench rc=$?" >> $OUT
grep -v amdgpu.ids $OUT; python3 -c "
import json; d=json.load(open('gpurun_out/bench_quick.json')); print({k:d[k] for k in ('value','ms_per_step','n_gpus')}); print(d['config']['launch']); print(d['roofline']['frac'], d['roofline']['launch_ms'])"
