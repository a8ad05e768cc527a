mkdir -p gpurun_out/cb
timeout 900 python -m pytest tests/test_gpu_fft_batch.py tests/test_gpu_baseline_parity.py -x -q -m gpu -k "fft_batch or batched or c6_shaped" -s > gpurun_out/cb/tests.log 2>&1; grep -n "^E  \|c6-shaped\|passed\|failed" gpurun_out/cb/tests.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_fit.py -x -q -m gpu -k "calibration or upsampling" 2>&1 | tail -2
for v in 1 0; do
JD_FFT_BATCH=$v timeout 300 python bench.py --config c6 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c6 batch=$v', d['ms_per_step'], d['host_enqueue_ms_per_step'], d['batched'], d['kernel_ms_per_step'])"
done
