#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r5d.log
: > $OUT
UAVTRACK_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --gather-transitions 4096 --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_2rank_tr.json 2> gpurun_out/bench_2rank_tr.err
echo "2 rank transitions (OMP_NUM_THREADS=1) rc=$?" >> $OUT
UAVTRACK_BENCH_ONE_GPU=1 OMP_NUM_THREADS=64 timeout -k 10 300 python3 bench.py --gpus 2 --backend gloo --gather-transitions 4096 --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_2rank_tr64.json 2> gpurun_out/bench_2rank_tr64.err
echo "2 rank transitions (OMP_NUM_THREADS=64) rc=$?" >> $OUT
timeout -k 10 300 python3 bench.py --gpus 1 --gather-transitions 4096 --no-extras --no-other-configs --no-cpu-baseline > gpurun_out/bench_1rank_tr.json 2> gpurun_out/bench_1rank_tr.err
echo "1 rank transitions rc=$?" >> $OUT
python3 - >> $OUT <<'PY'
import json
for f in ("bench_2rank_tr", "bench_2rank_tr64", "bench_1rank_tr"):
    try:
        d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
        print(f, {k: d[k] for k in ("value", "ms_per_step", "ms_per_step_per_rank", "gather")})
    except Exception as e:
        print(f, "unreadable", e)
PY
grep -v amdgpu.ids $OUT
