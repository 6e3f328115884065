#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5e.log
: > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5e_suite.log 2>&1
echo "suite rc=$?" >> $OUT; tail -4 gpurun_out/r5e_suite.log >> $OUT
timeout -k 10 500 python3 bench.py > gpurun_out/bench_r5e.json 2> gpurun_out/bench_r5e.err
echo "bench rc=$?" >> $OUT
python3 - >> $OUT <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r5e.json").read().strip().splitlines()[-1])
for c in d["configs"]:
    print(c)
PY
grep -v amdgpu.ids $OUT
