#!/bin/bash
# Device assembly of one translation unit: tools/asm_dump.sh gmm [extra flags] -> /tmp/asm/<name>.s
mkdir -p /tmp/asm
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form --cuda-device-only -S ${@:2} /root/repo/jolideco_amd/csrc/$1.hip -o /tmp/asm/$1.s 2>&1 | grep "error" | tail -5
