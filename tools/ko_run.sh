#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras "$@" > gpurun_out/bench_$name.json 2> gpurun_out/bench_$name.err; echo "$name rc=$?"; tail -1 gpurun_out/bench_$name.json | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('  value %.3f G  ms/step %.5f  roofline %s frac %.3f  avg_launch_ms %.3f' % (d['value']/1e9, d['ms_per_step'], r['bound'], r['frac'], r.get('avg_launch_ms', 0)))"; }
run mean --reward mean --steps 1000 --warmup 200
run pmi --reward pmi --steps 1000 --warmup 200
run pmi_dense --reward pmi --box 500 --steps 600 --warmup 200
run pmi_step --reward pmi --rollout 1 --steps 400 --warmup 100
run pmi_actor --reward pmi --policy actor --steps 1000 --warmup 200
run actor --policy actor --steps 1000 --warmup 200
run greedy --policy greedy --steps 1000 --warmup 200
run c4 --envs 8192 --n-uav 50 --m-targets 25 --dim 3 --steps 400 --warmup 200
run c4_actor --envs 8192 --n-uav 50 --m-targets 25 --dim 3 --policy actor --steps 400 --warmup 200
run sat --envs 65536 --steps 400 --warmup 200
run sat131k --envs 131072 --steps 200 --warmup 200
run sat_actor --envs 65536 --policy actor --steps 400 --warmup 200
