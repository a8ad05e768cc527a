#!/bin/bash
mkdir -p gpurun_out/r5k
python -m pytest tests/test_gpu_baseline_parity.py -x -q -m gpu -k "c6_shaped" -s > gpurun_out/r5k/t1.log 2>&1; echo "rc=$?" >> gpurun_out/r5k/t1.log
python -m pytest tests/test_gpu_fit.py tests/test_gpu_fft_batch.py tests/test_gpu_fft_native.py tests/test_gpu_edge_cases.py tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r5k/t2.log 2>&1; echo "rc=$?" >> gpurun_out/r5k/t2.log
python bench.py --config c6 > gpurun_out/r5k/c6.json 2> gpurun_out/r5k/c6.err
python bench.py --config e0102 --epochs 100 > gpurun_out/r5k/e0102_100.json 2> gpurun_out/r5k/e0102_100.err
python tools/gpu/small_fits.py > gpurun_out/r5k/small_fits.txt 2>&1
tail -n 4 gpurun_out/r5k/t1.log gpurun_out/r5k/t2.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5k/c6.json')); print('c6', d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline_c6']['frac'], {k:round(v['frac'],3) for k,v in d['roofline_c6']['launches'].items()})
d=json.load(open('gpurun_out/r5k/e0102_100.json')); print('e0102', d['value'], d['ms_per_step'], d['epoch_ms'], d['graph_policy'])
PY
grep "by-value\|graph   " gpurun_out/r5k/small_fits.txt
