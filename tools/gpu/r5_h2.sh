#!/bin/bash
O=gpurun_out/r5h2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shared_operator.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 15 $O/t.log
timeout 600 python tools/gpu/r5_h2.py 16 > $O/c5shared.txt 2> $O/c5shared.err; echo "rc=$?" >> $O/c5shared.txt; grep -v amdgpu.ids $O/c5shared.txt
timeout 600 python -m pytest tests/test_gpu_baseline_parity.py -x -q -m gpu -k "c5 or config5" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log; tail -n 3 $O/t2.log
