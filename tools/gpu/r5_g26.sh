#!/bin/bash
O=gpurun_out/r5ad; mkdir -p $O
python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_fit.py tests/test_gpu_fft_batch.py tests/test_gpu_edge_cases.py tests/test_gpu_graph.py -x -q -m gpu -k "not fullsize" > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 3 $O/t1.log
for io in 1 0 1 0; do JD_FFT_POOL_IO=$io python bench.py --config c6 > $O/c6_$io.json 2> $O/c6_$io.err; python - $O/c6_$io.json $io <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('c6 pool_io', sys.argv[2], round(d['value'],1), round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, round(d['roofline_c6']['frac'],3), {k: round(v['frac'],3) for k,v in d['roofline_c6']['launches'].items()})
PY
done
for io in 1 0; do JD_FFT_POOL_IO=$io python bench.py --config e0102 > $O/e0102_$io.json 2> $O/e0102_$io.err; python - $O/e0102_$io.json $io <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('e0102 pool_io', sys.argv[2], d['value'], d.get('ms_per_step'), d.get('graph_policy'))
PY
done
JD_FFT_POOL_IO=1 python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto"
