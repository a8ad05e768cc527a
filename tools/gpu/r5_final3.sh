#!/bin/bash
O=gpurun_out/r5final3; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$?"
python -m pytest tests -x -q -m gpu > $O/full.log 2>&1; echo "rc=$?" >> $O/full.log
tail -n 4 $O/full.log
python3 bench.py --steps 20 --warmup 5 > $O/c3_driver_flags.json 2> $O/c3.err; python - $O/c3_driver_flags.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('c3', round(d['value'],1), round(d['ms_per_step'],4), 'c6', round(d['c6_chandra_like']['value'],1), 'fft', round(d['fft_psf']['value'],1), 'roofline', round(d['roofline']['frac'],3), round(d['roofline_poisson']['frac'],3), 'clock', round(d['clock_mhz']))
PY
