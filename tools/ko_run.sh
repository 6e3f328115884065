#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/final_r02.log
: > $OUT
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" >> $OUT 2>&1
echo "smoke rc=$?" >> $OUT
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_k20.json 2> gpurun_out/bench_k20.err
echo "bench k20 rc=$?" >> $OUT
timeout -k 10 400 python3 bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "bench default rc=$?" >> $OUT
python3 - >> $OUT <<'PY'
import json
for f in ("bench_k20","bench_default"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, round(d["value"]/1e9,2),"G ms/step",round(d["ms_per_step"],5),"| roofline frac",round(r["frac"],3),"launch ms",round(r["avg_launch_ms"],4),"| cpu",round(d["cpu_baseline"]["value"]/1e6,1),"M", "| launch:",d["config"]["launch"][:90])
    for k in ("saturating_batch","closed_loop","per_step_launch"):
        if k in d: print("   ",k, json.dumps(d[k])[:300])
PY
grep -v amdgpu.ids $OUT
