#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/x6_r02.log
: > $OUT
timeout -k 10 600 python3 -m pytest tests -x -q -m gpu -k "pmi or baseline or auto_reset or fused_actor" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $OUT
tail -15 gpurun_out/pytest_gpu.log >> $OUT
timeout -k 10 300 python3 bench.py --reward pmi --steps 400 --warmup 200 --no-extras --no-cpu-baseline > gpurun_out/bench_pmi.json 2> gpurun_out/bench_pmi.err
echo "bench rc=$?" >> $OUT
UAVTRACK_PMI_FP32=1 timeout -k 10 300 python3 bench.py --reward pmi --steps 400 --warmup 200 --no-extras --no-cpu-baseline > gpurun_out/bench_pmi_fp32.json 2> gpurun_out/bench_pmi_fp32.err
echo "bench fp32 rc=$?" >> $OUT
python3 - >> $OUT <<'PY'
import json
for f in ("gpurun_out/bench_pmi.json","gpurun_out/bench_pmi_fp32.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d["value"]/1e9, "G", d["ms_per_step"], d.get("roofline",{}).get("avg_launch_ms"))
    except Exception as e: print(f, "ERR", e)
PY
grep -v amdgpu.ids $OUT
