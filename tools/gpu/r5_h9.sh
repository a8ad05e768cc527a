#!/bin/bash
O=gpurun_out/r5h9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fft_batch.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 3 $O/t.log
timeout 900 python -m pytest tests/test_gpu_baseline_parity.py tests/test_gpu_graph.py -x -q -m gpu -k "c6 or calibrat or factor or graph or replay" > $O/t2.log 2>&1; echo "rc=$?" >> $O/t2.log; tail -n 3 $O/t2.log
STEPS=30 bash tools/ab_libs.sh 3 c6 prev default > $O/c6.txt 2>&1; grep -v amdgpu.ids $O/c6.txt | cut -c1-330
for lib in prev default prev default; do if [ "$lib" = "default" ]; then unset JOLIDECO_HIP_LIBRARY; else export JOLIDECO_HIP_LIBRARY=jolideco_amd/libjolideco_hip_$lib.so; fi
SMALL_FITS_ONLY=256:uniform:graph timeout 200 python tools/gpu/small_fits.py 2> /dev/null | grep -v amdgpu | sed "s/^/$lib /"
timeout 300 python bench.py --config e0102 > $O/e0102_$lib.json 2> $O/e0102_$lib.err; python - $O/e0102_$lib.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('e0102', sys.argv[2], d['value'], d.get('ms_per_step'))
PY
done
