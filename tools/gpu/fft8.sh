mkdir -p gpurun_out/fft8; rm -f gpurun_out/fft8/ab.txt
timeout 900 python -m pytest tests/test_gpu_fft_native.py -x -q -m gpu > gpurun_out/fft8/tests.log 2>&1; tail -3 gpurun_out/fft8/tests.log
for r in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export JOLIDECO_HIP_LIBRARY=jolideco_amd/libjolideco_hip_old.so; else unset JOLIDECO_HIP_LIBRARY; fi
  JOLIDECO_CONV_METHOD=fft timeout 300 python tools/ab.py c3 1 30 -- $lib: 2>&1 | grep step >> gpurun_out/fft8/ab.txt
  timeout 300 python bench.py --config c6 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c6 $lib', d['ms_per_step'], d['kernel_ms_per_step'])" >> gpurun_out/fft8/ab.txt
done; done
cat gpurun_out/fft8/ab.txt
