#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5g.log
: > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/r5g_suite.log 2>&1
echo "suite rc=$?" >> $OUT; tail -4 gpurun_out/r5g_suite.log >> $OUT
timeout -k 10 300 python3 - >> $OUT 2>&1 <<'PY'
import sys
sys.path[:0] = [".", "marl-uavs-targets-tracking_amd"]
import uavtrack, bench
for c in bench.compat_leg(uavtrack)["shapes"]:
    print(c["cfg"], "%.1f us per step" % c["us_per_step"], ["%.1f" % x for x in c["us_per_step_all_runs"]], "x%.0f" % c["speedup_vs_reference"])
PY
grep -v amdgpu.ids $OUT
