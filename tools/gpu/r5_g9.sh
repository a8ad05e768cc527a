#!/bin/bash
# the prior's first phase beside the likelihood: bit-identity tests, then on / off on c3, c4, c6 and the small fits
O=gpurun_out/r5m; mkdir -p $O
python -m pytest tests/test_gpu_graph.py tests/test_gpu_fit.py -x -q -m gpu > $O/t1.log 2>&1; echo "rc=$?" >> $O/t1.log
tail -n 4 $O/t1.log
for cfg in c3 c4 c6; do
  for ov in 1 0 1 0; do
    JOLIDECO_PRIOR_OVERLAP=$ov python bench.py --config $cfg > $O/${cfg}_ov${ov}.json 2> $O/${cfg}_ov${ov}.err
    python - $O/${cfg}_ov${ov}.json $cfg $ov <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); print(sys.argv[2], 'overlap', sys.argv[3], d['value'], d['ms_per_step'])
except Exception as e:
    print(sys.argv[2], sys.argv[3], 'failed', e)
PY
  done
done
for ov in 1 0; do JOLIDECO_PRIOR_OVERLAP=$ov python tools/gpu/small_fits.py > $O/small_ov$ov.txt 2>&1; tail -n 8 $O/small_ov$ov.txt; done
