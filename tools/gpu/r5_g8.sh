#!/bin/bash
mkdir -p gpurun_out/r5l
python -m pytest tests/test_gpu_baseline_parity.py -x -q -m gpu -k "c6_shaped" > gpurun_out/r5l/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5l/t1.log
python -m pytest tests/test_gpu_fit.py tests/test_gpu_fft_batch.py tests/test_gpu_kernels.py -x -q -m gpu -k "upsampl or calib or pooled or batch" > gpurun_out/r5l/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5l/t2.log
python bench.py --config c6 > gpurun_out/r5l/c6.json 2> gpurun_out/r5l/c6.err
tail -n 3 gpurun_out/r5l/t1.log gpurun_out/r5l/t2.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5l/c6.json')); print('c6', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline_c6']['frac'], {k:round(v['frac'],3) for k,v in d['roofline_c6']['launches'].items()})
PY
