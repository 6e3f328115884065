#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/chk_r02.log
: > $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT
tail -5 gpurun_out/pytest_gpu.log >> $OUT
timeout -k 10 600 python3 tests/fuzz_api.py 80 > gpurun_out/fuzz.log 2>&1
echo "fuzz rc=$? $(tail -1 gpurun_out/fuzz.log)" >> $OUT
for i in 1 2; do
timeout -k 10 300 python3 bench.py --reward pmi --steps 400 --warmup 200 --no-extras --no-cpu-baseline > gpurun_out/bench_pmi.json 2> gpurun_out/bench_pmi.err
python3 - >> $OUT <<'PY'
import json
d=json.loads(open("gpurun_out/bench_pmi.json").read().strip().splitlines()[-1])
print(d["value"]/1e9, "G", d["roofline"]["avg_launch_ms"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
done
cat $OUT
