#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/pmi_r02.log
: > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT; tail -5 gpurun_out/pytest_gpu.log >> $OUT
timeout -k 10 300 python3 bench.py --reward pmi --steps 1000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/bench_pmi.json 2> gpurun_out/bench_pmi.err; echo "bench rc=$?" >> $OUT
grep -v amdgpu.ids $OUT; python3 -c "
import json; d=json.load(open('gpurun_out/bench_pmi.json')); print({k:d[k] for k in ('value','ms_per_step','n_gpus')}); print(d['roofline'])"
