#!/bin/bash
# round 5, second GPU call: in-kernel stamps of short launches, the driver's bench line, N = 1 vs the 2-rank one-GPU rehearsal
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5b.log
: > $OUT
timeout -k 10 200 python3 tools/experiments/stamp_run.py 4096 > gpurun_out/stamps.log 2>&1
echo "stamps rc=$?" >> $OUT; grep -v amdgpu.ids gpurun_out/stamps.log >> $OUT
timeout -k 10 200 python3 tools/experiments/stamp_run.py 1024 >> gpurun_out/stamps.log 2>&1
tail -4 gpurun_out/stamps.log >> $OUT
timeout -k 10 120 python3 tools/launch_overhead.py --T 1 >> $OUT 2>&1
timeout -k 10 300 python3 -m pytest tests/test_hip_parity.py -x -q -m gpu -k "bench_contract" > gpurun_out/r5b_contract.log 2>&1
echo "contract rc=$?" >> $OUT; tail -5 gpurun_out/r5b_contract.log >> $OUT
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_1rank_k20.json 2> gpurun_out/bench_1rank_k20.err
echo "1 rank k20 rc=$?" >> $OUT
UAVTRACK_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --steps 20 --warmup 5 --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_2rank_k20.json 2> gpurun_out/bench_2rank_k20.err
echo "2 rank k20 rc=$?" >> $OUT
timeout -k 10 300 python3 bench.py --gpus 1 --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_1rank.json 2> gpurun_out/bench_1rank.err
echo "1 rank default rc=$?" >> $OUT
UAVTRACK_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo "2 rank default rc=$?" >> $OUT
UAVTRACK_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --gather-transitions 4096 --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_2rank_tr.json 2> gpurun_out/bench_2rank_tr.err
echo "2 rank transitions rc=$?" >> $OUT
python3 - >> $OUT <<'PY'
import json
for f in ("bench_1rank_k20", "bench_2rank_k20", "bench_1rank", "bench_2rank", "bench_2rank_tr"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: d[k] for k in ("value", "ms_per_step", "ms_per_step_per_rank", "gather")})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -v amdgpu.ids $OUT
