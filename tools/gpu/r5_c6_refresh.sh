#!/bin/bash
set -e
O=gpurun_out/r5c6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 bench.py --config c6 > $O/c6_n1_bench.json 2> $O/c6.err
python3 bench.py --config e0102 > $O/e0102_bench.json 2> $O/e0102.err
export JOLIDECO_GRAPH=0 JOLIDECO_STEP_SCALARS=host
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c6 -o c6 -- python3 bench.py --config c6 --steps 20 --warmup 5 --no-cpu-baseline --no-general-psf > $O/prof_c6.log 2>&1
cp $(find $O/prof_c6 -name "*kernel_stats.csv" | sort | sed -n 1p) $O/c6_n1_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c6/c6_n1_bench.json').read().strip().splitlines()[-1]); print('c6', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline_c6']['frac'],3), {k: round(v['frac'],3) for k,v in d['roofline_c6']['launches'].items()})
d=json.loads(open('gpurun_out/r5c6/e0102_bench.json').read().strip().splitlines()[-1]); print('e0102', d['value'], d['ms_per_step'])
PY
