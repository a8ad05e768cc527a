#!/bin/bash
O=gpurun_out/r5h1; mkdir -p $O
for c in c3 c5 c6 c4 c2; do timeout 300 python tools/gpu/r5_h1.py $c 5 40 > $O/$c.txt 2> $O/$c.err; echo "rc=$?" >> $O/$c.txt; grep -v amdgpu.ids $O/$c.txt; done
