mkdir -p gpurun_out/cal1
timeout 1200 python -m pytest tests/test_gpu_fit.py tests/test_gpu_distributed.py tests/test_gpu_random_parity.py -x -q -k "calibrat or random" > gpurun_out/cal1/tests.txt 2>&1
tail -4 gpurun_out/cal1/tests.txt
python bench.py --config c6 > gpurun_out/cal1/c6.json 2> gpurun_out/cal1/c6.err
python3 -c "
import json; d=json.loads(open('gpurun_out/cal1/c6.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['kernel_ms_per_step'])"
