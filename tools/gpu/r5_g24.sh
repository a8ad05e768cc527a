#!/bin/bash
O=gpurun_out/r5ab; mkdir -p $O
python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fit.py tests/test_gpu_fft_batch.py tests/test_gpu_edge_cases.py tests/test_gpu_graph.py tests/test_gpu_fft_native.py -x -q -m gpu > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 3 $O/t1.log
for i in 1 2; do python bench.py --config c6 > $O/c6_$i.json 2> $O/c6_$i.err; python - $O/c6_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('c6', round(d['value'],1), round(d['ms_per_step'],4), d['kernel_ms_per_step'], round(d['roofline_c6']['frac'],3), {k: round(v['frac'],3) for k,v in d['roofline_c6']['launches'].items()})
PY
done
python bench.py --config e0102 > $O/e0102.json 2> $O/e0102.err; python - $O/e0102.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('e0102', d['value'], d['unit'], d.get('ms_per_step'), d.get('graph_policy'))
PY
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto\|graph  "
