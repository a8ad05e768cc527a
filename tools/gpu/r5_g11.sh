#!/bin/bash
# transposed shift with the gradient in registers over the datasets: parity tests, then the small fits and c6
O=gpurun_out/r5o; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fft_batch.py tests/test_gpu_edge_cases.py tests/test_gpu_graph.py -x -q -m gpu > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 4 $O/t1.log
python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fit.py -x -q -m gpu -k "calib or c6 or shift" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log
tail -n 4 $O/t2.log
python tools/gpu/small_fits.py > $O/small.txt 2>&1; cat $O/small.txt | grep flux
for i in 1 2; do
python bench.py --config c6 > $O/c6_$i.json 2> $O/c6_$i.err
python - $O/c6_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('c6', d['value'], d['ms_per_step'], d['kernel_ms_per_step'])
PY
done
