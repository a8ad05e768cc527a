#!/bin/bash
O=gpurun_out/r5ac; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q -m gpu -s > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
grep -a "auto policy\|passed\|failed\|rc=" $O/t1.log | tail -n 6
for i in 1 2; do python bench.py --config e0102 > $O/e0102_$i.json 2> $O/e0102_$i.err; python - $O/e0102_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print('e0102', d['value'], d['unit'], d.get('ms_per_step'), d.get('graph_policy'))
PY
done
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto\|graph  "
for cfg in c1 c2; do python bench.py --config $cfg --no-cpu-baseline > $O/$cfg.json 2> $O/$cfg.err; python - $O/$cfg.json $cfg <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d.get('ms_per_step'), d.get('graph_policy'))
PY
done
