#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/ko_r02.log
: > $OUT
for v in k_base k_lds k_bar k_gat k_rec k_all k_base; do
UAVTRACK_LIB=build_variants/$v.so timeout -k 10 300 python3 bench.py --reward pmi --steps 400 --warmup 200 --no-extras --no-cpu-baseline > gpurun_out/bench_pmi.json 2> gpurun_out/bench_pmi.err
python3 - $v >> $OUT <<'PY'
import json,sys
d=json.loads(open("gpurun_out/bench_pmi.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["value"]/1e9, "G", d["roofline"]["avg_launch_ms"], d["roofline"]["pairs_scored"])
PY
done
cat $OUT
