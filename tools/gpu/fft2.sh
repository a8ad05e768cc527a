mkdir -p gpurun_out/fft2
timeout 600 python -m pytest tests/test_gpu_fft_native.py -x -q > gpurun_out/fft2/tests.txt 2>&1
tail -3 gpurun_out/fft2/tests.txt
for v in 8 4 2; do
  echo "== columns per block $v" >> gpurun_out/fft2/ab.txt
  JD_FFT_NATIVE=$v JOLIDECO_CONV_METHOD=fft timeout 600 python3 tools/ab.py c3 3 20 -- native: >> gpurun_out/fft2/ab.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/fft2/ab.txt | cut -c1-330
