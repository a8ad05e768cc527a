#!/bin/bash
# Alternate full processes over library variants (compile-time switches): tools/ab_libs.sh ROUNDS CONFIG lib1 lib2 ...
# (lib = suffix of jolideco_amd/libjolideco_hip_<suffix>.so, "default" = the in-tree build)
R=$1; CFG=$2; shift 2
for r in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "default" ]; then unset JOLIDECO_HIP_LIBRARY; else export JOLIDECO_HIP_LIBRARY=jolideco_amd/libjolideco_hip_$lib.so; fi
    python tools/ab.py $CFG 1 ${STEPS:-40} -- $lib: > /tmp/ab_one.log 2>&1 || { echo "$lib FAILED:"; tail -n 5 /tmp/ab_one.log; }
    grep " step " /tmp/ab_one.log || true
  done
done
