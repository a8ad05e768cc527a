#!/bin/bash
mkdir -p gpurun_out/r5d
python -m pytest tests/test_gpu_fft_native.py tests/test_gpu_fft_batch.py tests/test_gpu_mixed_psf.py -x -q -m gpu > gpurun_out/r5d/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5d/t1.log
python -m pytest tests/test_gpu_baseline_parity.py -x -q -m gpu -k "c6_shaped" > gpurun_out/r5d/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5d/t2.log
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fit.py tests/test_gpu_distributed.py tests/test_gpu_edge_cases.py -x -q -m gpu > gpurun_out/r5d/t3.log 2>&1; echo "rc=$?" >> gpurun_out/r5d/t3.log
python tools/ab.py c6 2 20 -- seq: b3:JD_FFT_BATCH=3 all:JD_FFT_BATCH=2 cb4:JD_FFT_NATIVE=4 > gpurun_out/r5d/ab_c6_opts.txt 2>&1
STEPS=20 bash tools/ab_libs.sh 2 c6 pf0 default > gpurun_out/r5d/ab_c6_pf.txt 2>&1
tail -n 3 gpurun_out/r5d/t1.log gpurun_out/r5d/t2.log gpurun_out/r5d/t3.log
grep " step " gpurun_out/r5d/ab_c6_opts.txt | cut -c1-450; cut -c1-450 gpurun_out/r5d/ab_c6_pf.txt
