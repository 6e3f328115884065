#!/bin/bash
# round 5, first GPU call: the new tests first (fail fast), then the whole suite, the overlap probe, the default bench line
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5a.log
: > $OUT
timeout -k 10 500 python3 -m pytest tests/test_hip_round5.py -x -q -m gpu > gpurun_out/r5a_new.log 2>&1
echo "new tests rc=$?" >> $OUT; tail -15 gpurun_out/r5a_new.log >> $OUT
timeout -k 10 240 python3 tools/experiments/overlap_probe.py > gpurun_out/overlap_probe.log 2>&1
echo "probe rc=$?" >> $OUT; grep -v amdgpu.ids gpurun_out/overlap_probe.log >> $OUT
UAVTRACK_TEST_REPORT=1 timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -s > gpurun_out/r5a_suite.log 2>&1
echo "suite rc=$?" >> $OUT; tail -6 gpurun_out/r5a_suite.log >> $OUT
grep "census\|knife-edge\] golden" gpurun_out/r5a_suite.log | tail -12 >> $OUT
timeout -k 10 500 python3 bench.py > gpurun_out/bench_r5a.json 2> gpurun_out/bench_r5a.err
echo "bench rc=$?" >> $OUT
grep -v amdgpu.ids $OUT
