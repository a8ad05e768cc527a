set -x
mkdir -p gpurun_out/fft8
timeout 900 python -m pytest tests/test_gpu_fft_native.py -x -q -m gpu > gpurun_out/fft8/tests.log 2>&1; tail -3 gpurun_out/fft8/tests.log
for r in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export JOLIDECO_HIP_LIBRARY=jolideco_amd/libjolideco_hip_old.so; else unset JOLIDECO_HIP_LIBRARY; fi
  JOLIDECO_CONV_METHOD=fft timeout 300 python tools/ab.py c3 1 30 -- $lib: 2>&1 | grep step >> gpurun_out/fft8/ab.txt
  timeout 300 python tools/ab.py c6 1 10 -- c6$lib: 2>&1 | grep step >> gpurun_out/fft8/ab.txt
done; done
cat gpurun_out/fft8/ab.txt
