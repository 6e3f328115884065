#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/chk_r02.log
: > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT
tail -5 gpurun_out/pytest_gpu.log >> $OUT
timeout -k 10 600 python3 tests/fuzz_api.py 60 > gpurun_out/fuzz.log 2>&1
echo "fuzz rc=$? $(tail -1 gpurun_out/fuzz.log)" >> $OUT
timeout -k 10 300 python3 bench.py --reward pmi --no-cpu-baseline > gpurun_out/bench_pmi.json 2> gpurun_out/bench_pmi.err
echo "bench pmi rc=$?" >> $OUT
tail -c 1500 gpurun_out/bench_pmi.json >> $OUT
cat $OUT
