#!/bin/bash
O=gpurun_out/r5t; mkdir -p $O
python -m pytest tests/test_gpu_graph.py tests/test_gpu_fit.py -x -q -m gpu > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 3 $O/t1.log
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto\|graph  "
python bench.py --config e0102 > $O/e0102.json 2> $O/e0102.err; python - $O/e0102.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('e0102', d['value'], d['unit'], d.get('ms_per_step'), d.get('graph_policy'))
PY
