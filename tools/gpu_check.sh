#!/bin/bash
# One gpurun call's worth of validation on the GPU box:
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/gpu_check.sh'
# full -m gpu suite, smoke(), the driver's bench invocation, a slice of the API fuzzer.  Results under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/gpu_check.log
: > $OUT
fail=0
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
rc=$?; [ $rc -ne 0 ] && fail=1; echo "pytest rc=$rc" >> $OUT
tail -4 gpurun_out/pytest_gpu.log >> $OUT
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" >> $OUT 2>&1
rc=$?; [ $rc -ne 0 ] && fail=1; echo "smoke rc=$rc" >> $OUT
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_k20.json 2> gpurun_out/bench_k20.err
rc=$?; [ $rc -ne 0 ] && fail=1; echo "bench rc=$rc" >> $OUT
timeout -k 10 600 python3 tests/fuzz_api.py 40 > gpurun_out/fuzz.log 2>&1
rc=$?; [ $rc -ne 0 ] && fail=1; echo "fuzz rc=$rc $(tail -1 gpurun_out/fuzz.log)" >> $OUT
timeout -k 10 600 python3 tests/soak.py > gpurun_out/soak.log 2>&1
rc=$?; [ $rc -ne 0 ] && fail=1; echo "soak rc=$rc $(tail -2 gpurun_out/soak.log | head -1)" >> $OUT
python3 tools/perf_floor.py gpurun_out/bench_k20.json >> $OUT 2>&1
grep -v amdgpu.ids $OUT
exit $fail
