#!/bin/bash
O=gpurun_out/r5h8; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fft_native.py tests/test_gpu_fft_batch.py tests/test_gpu_kernels.py tests/test_gpu_edge_cases.py tests/test_gpu_fit.py tests/test_gpu_mixed_psf.py -x -q -m gpu > $O/t.log 2>&1; echo "rc=$?" >> $O/t.log; tail -n 3 $O/t.log
STEPS=30 bash tools/ab_libs.sh 2 c6 base default > $O/c6.txt 2>&1; grep -v amdgpu.ids $O/c6.txt | cut -c1-330
STEPS=40 bash tools/ab_libs.sh 2 c3fft base default > $O/c3fft.txt 2>&1; grep -v amdgpu.ids $O/c3fft.txt | cut -c1-330
STEPS=100 bash tools/ab_libs.sh 2 c2 base default > $O/c2.txt 2>&1; grep -v amdgpu.ids $O/c2.txt | cut -c1-330
for lib in base default; do if [ "$lib" = "default" ]; then unset JOLIDECO_HIP_LIBRARY; else export JOLIDECO_HIP_LIBRARY=jolideco_amd/libjolideco_hip_$lib.so; fi
timeout 300 python bench.py --config e0102 > $O/e0102_$lib.json 2> $O/e0102_$lib.err; python - $O/e0102_$lib.json $lib <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('e0102', sys.argv[2], d['value'], d.get('ms_per_step'))
PY
done
