#!/bin/bash
O=gpurun_out/r5c5; mkdir -p $O
python3 bench.py --config c5 --steps 100 --warmup 10 > $O/c5_n1_bench.json 2> $O/c5.err; echo "rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c5/c5_n1_bench.json').read().strip().splitlines()[-1]); print('c5', round(d['value'],1), round(d['ms_per_step'],4), 'shared_psf', d.get('shared_psf'))
PY
