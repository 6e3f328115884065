#!/bin/bash
# ALLFLAGS="-D..." build_variant.sh NAME [extra hipcc flags for step_kernel.hip...]  (ALLFLAGS reach every file) -> build_variants/NAME.so (A/B timing with tools/sweep.py --lib)
# The unchanged translation units are taken from the main build (csrc/build/*.o) unless ALLFLAGS is set.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
SRC=$ROOT/marl-uavs-targets-tracking_amd/csrc
OUT=$ROOT/build_variants; mkdir -p $OUT/obj_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=off -I$ROOT/include -I$SRC -Wno-unused-function -Wno-pass-failed"
for f in api reset_kernel pmi_kernel policy_kernel; do
  if [ -z "$ALLFLAGS" ] && [ -f $SRC/build/$f.o ] && [ -z "$REBUILD_ALL" ]; then cp $SRC/build/$f.o $OUT/obj_$NAME/$f.o
  else EXTRA=""; [ $f = pmi_kernel ] && EXTRA="-fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 $PMIFLAGS"; [ $f = policy_kernel ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1"
       /opt/rocm/bin/hipcc $FLAGS $ALLFLAGS $EXTRA -c $SRC/$f.hip -o $OUT/obj_$NAME/$f.o & fi
done
if [ -n "$SKIP_STEP" ] && [ -f $SRC/build/step_kernel.o ]; then cp $SRC/build/step_kernel.o $OUT/obj_$NAME/step_kernel.o
else /opt/rocm/bin/hipcc $FLAGS $ALLFLAGS -fno-convergent-functions -fno-slp-vectorize -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c $SRC/step_kernel.hip -o $OUT/obj_$NAME/step_kernel.o
fi
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OUT/obj_$NAME/*.o
echo built $OUT/$NAME.so
