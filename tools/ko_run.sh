#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/x6_r02.log
: > $OUT
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "pmi or baseline or auto_reset or fused_actor or fuzz or example" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT
tail -4 gpurun_out/pytest_gpu.log >> $OUT
for w in 0 128 64; do
UAVTRACK_WGS=$w timeout -k 10 300 python3 bench.py --reward pmi --steps 400 --warmup 200 --no-extras --no-cpu-baseline > gpurun_out/bench_pmi.json 2> gpurun_out/bench_pmi.err
python3 - $w >> $OUT <<'PY'
import json,sys
d=json.loads(open("gpurun_out/bench_pmi.json").read().strip().splitlines()[-1])
print("wgs",sys.argv[1], d["value"]/1e9, "G", d["roofline"]["avg_launch_ms"])
PY
done
timeout -k 10 300 python3 bench.py --reward pmi --rollout 1 --steps 400 --warmup 100 --no-extras --no-cpu-baseline > gpurun_out/bench_pmi1.json 2> gpurun_out/bench_pmi1.err
python3 - >> $OUT <<'PY'
import json
d=json.loads(open("gpurun_out/bench_pmi1.json").read().strip().splitlines()[-1])
print("T=1:", d["value"]/1e9, "G", d["ms_per_step"])
PY
cat $OUT
