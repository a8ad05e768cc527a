#!/bin/bash
# round 5, FFT kernels: packed complex arithmetic + 576-thread rows at 4608 + linear LDS indices, against the round-4 build
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_fft_native.py tests/test_gpu_fft_batch.py -x -q -m gpu > gpurun_out/r5b/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/t1.log
python -m pytest tests/test_gpu_baseline_parity.py -x -q -m gpu -k "c6_shaped" -s > gpurun_out/r5b/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/t2.log
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_fit.py -x -q -m gpu -k "fft or calib or upsampl or conv" > gpurun_out/r5b/t3.log 2>&1; echo "rc=$?" >> gpurun_out/r5b/t3.log
bash tools/ab_libs.sh 2 c6 base default > gpurun_out/r5b/ab_c6.txt 2>&1
bash tools/ab_libs.sh 2 c3fft base default > gpurun_out/r5b/ab_c3fft.txt 2>&1
tail -n 3 gpurun_out/r5b/t1.log gpurun_out/r5b/t2.log gpurun_out/r5b/t3.log
cut -c1-400 gpurun_out/r5b/ab_c6.txt gpurun_out/r5b/ab_c3fft.txt
