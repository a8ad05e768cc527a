mkdir -p gpurun_out/fft5
for lib in base lin base lin; do
  if [ $lib = lin ]; then export JOLIDECO_HIP_LIBRARY=$PWD/jolideco_amd/libjolideco_hip_lin.so; else unset JOLIDECO_HIP_LIBRARY; fi
  echo "== $lib" >> gpurun_out/fft5/ab.txt
  JOLIDECO_CONV_METHOD=fft timeout 600 python3 tools/ab.py c3 3 20 -- fft: >> gpurun_out/fft5/ab.txt 2>&1
done
export JOLIDECO_HIP_LIBRARY=$PWD/jolideco_amd/libjolideco_hip_lin.so
timeout 600 python -m pytest tests/test_gpu_fft_native.py -x -q 2>&1 | tail -2
grep -v amdgpu.ids gpurun_out/fft5/ab.txt | cut -c1-330
