#!/bin/bash
O=gpurun_out/r5s; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fft_batch.py tests/test_gpu_edge_cases.py tests/test_gpu_graph.py -x -q -m gpu > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 3 $O/t1.log
python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fit.py -x -q -m gpu -k "calib or c6 or shift" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log
tail -n 3 $O/t2.log
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto\|graph  "
