#!/bin/bash
O=gpurun_out/r5q; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log
tail -n 5 $O/full.log
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt
for cfg in e0102 c3 c6; do python bench.py --config $cfg > $O/$cfg.json 2> $O/$cfg.err; python - $O/$cfg.json $cfg <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['unit'], d.get('ms_per_step'), d.get('graph_policy'))
PY
done
