#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
OUT=gpurun_out/st_r02.log
: > $OUT
for v in st_1 st_2 st_3; do
UAVTRACK_LIB=build_variants/$v.so timeout -k 10 300 python3 bench.py --reward pmi --steps 400 --warmup 200 --no-extras --no-cpu-baseline > gpurun_out/bench_pmi.json 2> gpurun_out/bench_pmi.err
python3 - $v >> $OUT <<'PY'
import json,sys
d=json.loads(open("gpurun_out/bench_pmi.json").read().strip().splitlines()[-1])
v=int(d["roofline"]["pairs_scored"])
tiles=v>>44; cyc=v&((1<<44)-1)
print(sys.argv[1], "launch ms", d["roofline"]["avg_launch_ms"], "tiles", tiles, "cycles", cyc, "per tile", cyc/max(tiles,1))
PY
done
cat $OUT
