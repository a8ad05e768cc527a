#!/bin/bash
O=gpurun_out/r5aa; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fft_batch.py tests/test_gpu_edge_cases.py tests/test_gpu_graph.py -x -q -m gpu > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 3 $O/t1.log
python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fit.py -x -q -m gpu -k "calib or c6 or shift" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log
tail -n 3 $O/t2.log
for i in 1 2; do python bench.py --config c6 > $O/c6_$i.json 2> $O/c6_$i.err; python - $O/c6_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('c6', round(d['value'],1), round(d['ms_per_step'],4), d['kernel_ms_per_step'].get('shift'), round(d['roofline_c6']['frac'],3), round(d['roofline_c6']['launches']['shift']['frac'],3))
PY
done
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto\|graph  "
