#!/bin/bash
# Build a variant of libjolideco_hip.so with extra compiler flags (A/B experiments on the GPU box):
#   tools/build_variant.sh NAME "-DJD_SCREEN_LDS_CONSTS=0 ..."   ->  jolideco_amd/libjolideco_hip_NAME.so
# Select it at run time with JOLIDECO_HIP_LIBRARY=jolideco_amd/libjolideco_hip_NAME.so
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BUILD=/tmp/jdvar_$NAME
mkdir -p $BUILD
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form $EXTRA"
pids=()
for f in elementwise fftconv fftnative directconv sepconv walkconv shift gmm profile options; do
  /opt/rocm/bin/hipcc $FLAGS -c $ROOT/jolideco_amd/csrc/$f.hip -o $BUILD/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 $BUILD/*.o -shared -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -o $ROOT/jolideco_amd/libjolideco_hip_$NAME.so
echo built $ROOT/jolideco_amd/libjolideco_hip_$NAME.so
