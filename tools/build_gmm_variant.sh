#!/bin/bash
# A variant of the library that differs in gmm.hip's compile-time switches only (the other objects are the in-tree build's):
#   tools/build_gmm_variant.sh NAME "-DJD_SCREEN_SCHED_NV=7"  ->  jolideco_amd/libjolideco_hip_NAME.so
set -e
NAME=$1; EXTRA=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/jolideco_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-result -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form $EXTRA -c $C/gmm.hip -o /tmp/gmm_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 $C/elementwise.o $C/fftconv.o $C/fftnative.o $C/directconv.o $C/sepconv.o $C/walkconv.o $C/shift.o /tmp/gmm_$NAME.o $C/profile.o $C/options.o -shared -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib -o $ROOT/jolideco_amd/libjolideco_hip_$NAME.so
echo built libjolideco_hip_$NAME.so
