#!/bin/bash
cd $GRAFT_REPO_ROOT
PROFILE_MFMA=1 bash tools/profile_gpu.sh r02pmi --reward pmi --steps 1000 --warmup 200 --no-cpu-baseline --no-extras > gpurun_out/prof_r02pmi.log 2>&1
echo "profile rc=$?"
tail -3 gpurun_out/prof_r02pmi.log
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 bench.py --reward pmi > gpurun_out/bench_pmi_full.json 2> gpurun_out/bench_pmi_full.err
echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --reward pmi --box 500 --no-cpu-baseline > gpurun_out/bench_pmi_dense.json 2> gpurun_out/bench_pmi_dense.err
echo "bench dense rc=$?"
timeout -k 10 300 python3 bench.py --reward pmi --policy actor --no-cpu-baseline > gpurun_out/bench_pmi_actor.json 2> gpurun_out/bench_pmi_actor.err
echo "bench actor rc=$?"
