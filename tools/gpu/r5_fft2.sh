#!/bin/bash
mkdir -p gpurun_out/r5c
STEPS=20 bash tools/ab_libs.sh 2 c6 base default > gpurun_out/r5c/ab_c6.txt 2>&1
STEPS=40 bash tools/ab_libs.sh 2 c3fft base default > gpurun_out/r5c/ab_c3fft.txt 2>&1
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fft_batch.py tests/test_gpu_distributed.py -x -q -m gpu -k "calib or shift or every_rank" > gpurun_out/r5c/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5c/t1.log
tail -n 3 gpurun_out/r5c/t1.log
cut -c1-600 gpurun_out/r5c/ab_c6.txt gpurun_out/r5c/ab_c3fft.txt
