#!/bin/bash
cd "${GRAFT_REPO_ROOT}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/ko.log
: > $OUT
A="--envs 4096 --T 200 --reset --warm 150 --reps 20"
for lib in marl-uavs-targets-tracking_amd/uavtrack/libuavtrack.so build_variants/koA.so build_variants/koB.so; do
  UAVTRACK_LIB_OLDER_OK=1 timeout -k 10 200 python3 tools/sweep.py --lib $lib $A 2>&1 | grep "lib=" >> $OUT
done
timeout -k 10 200 python3 tools/sweep.py $A --wgs 64,128,256 2>&1 | grep "lib=" >> $OUT
echo "--- 2048 envs (1 wave per SIMD at most)" >> $OUT
timeout -k 10 200 python3 tools/sweep.py --envs 3072 --T 200 --reset --warm 150 --reps 20 2>&1 | grep "lib=" >> $OUT
timeout -k 10 200 python3 tools/sweep.py --envs 6144 --T 200 --reset --warm 150 --reps 20 2>&1 | grep "lib=" >> $OUT
cat $OUT
