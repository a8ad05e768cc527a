#!/bin/bash
O=gpurun_out/r5w; mkdir -p $O
python -m pytest tests/test_gpu_graph.py -x -q -m gpu -s > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
grep -a "auto policy\|passed\|failed\|rc=" $O/t1.log | tail -n 6
for cfg in c1 c2 c3 c4 c5 c6; do
  python bench.py --config $cfg --no-cpu-baseline > $O/$cfg.json 2> $O/$cfg.err
  python - $O/$cfg.json $cfg <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(f"{sys.argv[2]:6s} {d['value']:8.1f} it/s {d['ms_per_step']:.4f} ms  host {d.get('host_enqueue_ms_per_step')}  {d.get('graph_policy')}")
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
done
python tools/gpu/small_fits.py > $O/small.txt 2>&1; grep flux $O/small.txt | grep "auto"
