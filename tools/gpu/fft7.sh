mkdir -p gpurun_out/fft7
for lib in base loop looplin base loop looplin; do
  if [ $lib = base ]; then unset JOLIDECO_HIP_LIBRARY; else export JOLIDECO_HIP_LIBRARY=$PWD/jolideco_amd/libjolideco_hip_$lib.so; fi
  echo "== $lib" >> gpurun_out/fft7/ab.txt
  JOLIDECO_CONV_METHOD=fft timeout 600 python3 tools/ab.py c3 3 20 -- fft: >> gpurun_out/fft7/ab.txt 2>&1
done
grep -v amdgpu.ids gpurun_out/fft7/ab.txt | cut -c1-330
