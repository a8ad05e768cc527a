mkdir -p gpurun_out/fft9; rm -f gpurun_out/fft9/ab.txt
for r in 1 2; do for v in 1 4; do
  JD_FFT_NATIVE=$v timeout 300 python bench.py --config c6 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c6 cb=$v', d['ms_per_step'], d['kernel_ms_per_step'])" >> gpurun_out/fft9/ab.txt
done; done
cat gpurun_out/fft9/ab.txt
