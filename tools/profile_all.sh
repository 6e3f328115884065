#!/bin/bash
# Every rocprofv3 profile of a round in ONE gpurun call (GPU box):  bash tools/profile_all.sh r04
# then, back in the build container:  bash tools/profile_all.sh r04 --summarise   (writes profiles/<tag>*)
R=${1:-r04}
C="--no-cpu-baseline --no-extras --no-other-configs"
if [ "$2" = "--summarise" ]; then
  python3 tools/summarise_profile.py ${R}      gpurun_out/prof_${R}      4096x20x10_T200
  python3 tools/summarise_profile.py ${R}c4    gpurun_out/prof_${R}c4    8192x50x25_T200_3d
  python3 tools/summarise_profile.py ${R}pmi   gpurun_out/prof_${R}pmi   4096x20x10_T200_pmi128
  python3 tools/summarise_profile.py ${R}pmi64 gpurun_out/prof_${R}pmi64 4096x20x10_T200_pmi64
  python3 tools/summarise_profile.py ${R}sat   gpurun_out/prof_${R}sat   65536x20x10_T200
  python3 tools/summarise_profile.py ${R}actor gpurun_out/prof_${R}actor
  exit 0
fi
set -e
bash tools/profile_gpu.sh ${R}      --steps 1000 --warmup 200 $C > gpurun_out/prof_${R}.log 2>&1
bash tools/profile_gpu.sh ${R}c4    --envs 8192 --n-uav 50 --m-targets 25 --dim 3 --steps 600 --warmup 200 $C > gpurun_out/prof_${R}c4.log 2>&1
PROFILE_MFMA=1 bash tools/profile_gpu.sh ${R}pmi   --reward pmi --steps 1000 --warmup 200 $C > gpurun_out/prof_${R}pmi.log 2>&1
PROFILE_MFMA=1 bash tools/profile_gpu.sh ${R}pmi64 --reward pmi --pmi-hidden 64 --steps 1000 --warmup 200 $C > gpurun_out/prof_${R}pmi64.log 2>&1
bash tools/profile_gpu.sh ${R}sat   --envs 65536 --steps 400 --warmup 200 $C > gpurun_out/prof_${R}sat.log 2>&1
PROFILE_MFMA=1 bash tools/profile_gpu.sh ${R}actor --policy actor --steps 1000 --warmup 200 $C > gpurun_out/prof_${R}actor.log 2>&1
for f in gpurun_out/prof_${R}*.log; do tail -n 2 $f; done
