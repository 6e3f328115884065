#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/actor_ab.log
: > $OUT
timeout -k 10 400 python3 -m pytest tests -x -q -m gpu -k "actor or policy or closed" > gpurun_out/actor_tests.log 2>&1
echo "actor tests rc=$?" >> $OUT; tail -3 gpurun_out/actor_tests.log >> $OUT
for i in 1 2; do
timeout -k 10 300 python3 bench.py --policy actor --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_actor.json 2> gpurun_out/bench_actor.err
python3 - >> $OUT <<'PY'
import json
d = json.loads(open("gpurun_out/bench_actor.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("actor fused: value %.3f G, roofline leg kernel %.4f ms per 200 steps -> %.2f G" % (d["value"] / 1e9, r["kernel_avg_ms"], 4096 * 20 * 200 / r["kernel_avg_ms"] / 1e6))
PY
done
grep -v amdgpu.ids $OUT
