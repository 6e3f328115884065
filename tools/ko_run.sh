#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_gpu.sh r02 --steps 1000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/profile_r02.log 2>&1; echo "r02 rc=$?"
PROFILE_MFMA=1 bash tools/profile_gpu.sh r02pmi --reward pmi --steps 1000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/profile_r02pmi.log 2>&1; echo "r02pmi rc=$?"
bash tools/profile_gpu.sh r02c4 --envs 8192 --n-uav 50 --m-targets 25 --dim 3 --steps 400 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/profile_r02c4.log 2>&1; echo "r02c4 rc=$?"
bash tools/profile_gpu.sh r02sat --envs 65536 --steps 200 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/profile_r02sat.log 2>&1; echo "r02sat rc=$?"
for t in r02 r02pmi r02c4 r02sat; do tail -1 gpurun_out/prof_$t/trace_bench.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['roofline']['frac'], d['roofline'].get('avg_launch_ms'))"; done
