mkdir -p gpurun_out/c6t
timeout 1200 python -m pytest tests/test_gpu_baseline_parity.py -q -m gpu -k "c6_shaped" -s > gpurun_out/c6t/tests.log 2>&1; grep -n "^E  \|c6-shaped\|passed\|failed" gpurun_out/c6t/tests.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_fit.py tests/test_gpu_fft_native.py -x -q -m gpu > gpurun_out/c6t/tests2.log 2>&1; tail -3 gpurun_out/c6t/tests2.log
timeout 300 python bench.py --config c6 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c6 fused', d['ms_per_step'], d['kernel_ms_per_step'])"
