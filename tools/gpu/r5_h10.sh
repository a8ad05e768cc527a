#!/bin/bash
O=gpurun_out/r5h10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_kernels.py tests/test_gpu_fft_batch.py tests/test_gpu_shared_operator.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 3 $O/t.log
timeout 400 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fft_native.py -x -q -m gpu -k "c6 or calibrat or factor or odd or ragged or 131 or 2047" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log; tail -n 3 $O/t2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
