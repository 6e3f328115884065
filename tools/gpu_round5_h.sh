#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5h.log
: > $OUT
timeout -k 10 900 python3 tests/soak.py > gpurun_out/soak.log 2>&1
echo "soak rc=$?" >> $OUT; tail -9 gpurun_out/soak.log >> $OUT
timeout -k 10 600 python3 tests/fuzz_api.py 60 > gpurun_out/fuzz.log 2>&1
echo "fuzz rc=$?" >> $OUT; tail -2 gpurun_out/fuzz.log >> $OUT
timeout -k 10 300 python3 tests/soak_pmi.py > gpurun_out/soak_pmi.log 2>&1
echo "soak_pmi rc=$?" >> $OUT; tail -3 gpurun_out/soak_pmi.log >> $OUT
timeout -k 10 300 python3 tests/soak.py --fuzz 60 > gpurun_out/soak_fuzz.log 2>&1
echo "soak fuzz rc=$?" >> $OUT; tail -2 gpurun_out/soak_fuzz.log >> $OUT
grep -v amdgpu.ids $OUT
