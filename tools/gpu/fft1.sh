mkdir -p gpurun_out/fft1
timeout 600 python -m pytest tests/test_gpu_fft_native.py -x -q > gpurun_out/fft1/tests.txt 2>&1
tail -15 gpurun_out/fft1/tests.txt
JOLIDECO_CONV_METHOD=fft timeout 600 python3 tools/ab.py c3 3 20 -- native: rocfft:JD_FFT_NATIVE=0 > gpurun_out/fft1/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/fft1/ab.txt | cut -c1-400
