#!/bin/bash
# ALLFLAGS="-D..." build_variant.sh NAME [extra hipcc flags for step_kernel.hip...]  (ALLFLAGS reach every file) -> build_variants/NAME.so (A/B timing with tools/sweep.py --lib)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=$ROOT/marl-uavs-targets-tracking_amd/csrc
OUT=$ROOT/build_variants; mkdir -p $OUT/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -I$ROOT/include -I$SRC -Wno-unused-function -Wno-pass-failed"
for f in api reset_kernel pmi_kernel policy_kernel; do
  /opt/rocm/bin/hipcc $FLAGS $ALLFLAGS -c $SRC/$f.hip -o $OUT/obj_$NAME/$f.o &
done
/opt/rocm/bin/hipcc $FLAGS $ALLFLAGS -fno-convergent-functions -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c $SRC/step_kernel.hip -o $OUT/obj_$NAME/step_kernel.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/obj_$NAME/*.o
echo built $OUT/$NAME.so
