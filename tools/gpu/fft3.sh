mkdir -p gpurun_out/fft3
JOLIDECO_CONV_METHOD=fft timeout 600 python3 tools/ab.py c3 3 20 -- full: nofft:JD_FFT_DEBUG=1 nokhat:JD_FFT_DEBUG=2 neither:JD_FFT_DEBUG=3 > gpurun_out/fft3/ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/fft3/ab.txt | cut -c1-330
