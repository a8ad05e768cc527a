#!/bin/bash
O=gpurun_out/r5r; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q -m gpu -s > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
grep -a "auto policy\|passed\|failed\|rc=" $O/t1.log | tail -n 6
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto\|graph  \|by-value"
python bench.py --config c1 > $O/c1.json 2> $O/c1.err; python - $O/c1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('c1', d['value'], d['unit'], d.get('ms_per_step'), d.get('graph_policy'))
PY
