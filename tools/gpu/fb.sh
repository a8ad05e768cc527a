mkdir -p gpurun_out/fb
timeout 900 python -m pytest tests/test_gpu_fft_batch.py -x -q -m gpu > gpurun_out/fb/tests.log 2>&1; tail -15 gpurun_out/fb/tests.log | cut -c1-250
JOLIDECO_CONV_METHOD=fft timeout 600 python tools/ab.py c3 3 30 -- batched: loop:JD_FFT_BATCH=0 > gpurun_out/fb/ab.txt 2>&1
grep step gpurun_out/fb/ab.txt | cut -c1-330
