#!/bin/bash
cd "${GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out
timeout -k 10 1000 python3 tests/fuzz_api.py 400 > gpurun_out/fuzz400.log 2>&1
echo "fuzz rc=$? $(tail -1 gpurun_out/fuzz400.log)"
timeout -k 10 600 python3 tests/soak.py --fuzz 600 > gpurun_out/soak_fuzz600.log 2>&1
echo "soak fuzz rc=$? $(tail -1 gpurun_out/soak_fuzz600.log)"
timeout -k 10 600 python3 tests/soak.py --quick > gpurun_out/soak_quick.log 2>&1
echo "soak quick rc=$? $(tail -2 gpurun_out/soak_quick.log | head -1 | cut -c1-300)"
